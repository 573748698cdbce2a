for i in 1 2; do for l in musicfpaugment_amd/libmfpa_base.so musicfpaugment_amd/libmfpa.so; do echo "== $l"; python - <<PY 2>&1 | grep -v amdgpu
import sys
sys.argv=['x','--clips','128','--reps','5']
from musicfpaugment_amd import _lib
_lib.set_library_path("$l")
exec(open('tools/exp_upconv.py').read())
PY
done; done
