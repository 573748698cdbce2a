python tools/exp_conv.py --both --reps 5 2>/dev/null | grep -v amdgpu | sed 's/^/base /'
python tools/exp_conv.py --both --reps 5 --lib musicfpaugment_amd/libmfpa_pr.so 2>/dev/null | grep -v amdgpu | sed 's/^/prows/'
python tools/exp_conv.py --both --reps 5 2>/dev/null | grep -v amdgpu | sed 's/^/base /'
python tools/exp_conv.py --both --reps 5 --lib musicfpaugment_amd/libmfpa_pr.so 2>/dev/null | grep -v amdgpu | sed 's/^/prows/'
