import sys, json, subprocess
sys.path.insert(0, '.')
import torch
from musicfpaugment_amd import ops_train
import bench, argparse
# in-process A/B: USE_BF16_DZ off / on
def run(flag):
    ops_train.USE_BF16_DZ = flag
    sys.argv = ["bench.py", "--mode", "train", "--precision", "bf16", "--steps", "10", "--warmup", "3"]
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print("USE_BF16_DZ", flag, d["value"], d["ms_per_step"], d["config"]["loss_last"], d["roofline"]["kernel_ms_per_step"], flush=True)
for f in (False, True, False, True):
    run(f)
