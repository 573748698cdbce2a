#!/usr/bin/env python3
"""Demucs forward with the two LSTM layers pipelined on two streams at batch sizes above ops_demucs.PIPELINE_MAX_CLIPS:
usage: exp_demucs_pipeline.py <max_clips> <chunk> [bench.py arguments]   (runs bench.py's main with the two module constants set)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import musicfpaugment_amd.ops_demucs as D
D.PIPELINE_MAX_CLIPS = int(sys.argv[1])
D.LSTM_CHUNK = int(sys.argv[2])
sys.argv = ["bench.py"] + sys.argv[3:]
import bench
bench.main()
