# same-call A/B of tools/exp_upconv.py between library builds: ab_upconv_libs.sh <clips> <lib> <lib> ...   (alternating, two rounds)
CLIPS=${1:-128}; shift
LIBS=${@:-"musicfpaugment_amd/libmfpa_base.so musicfpaugment_amd/libmfpa.so"}
for i in 1 2; do for l in $LIBS; do echo "== $l"; python - <<PY 2>&1 | grep -v amdgpu
import sys
sys.argv=['x','--clips','$CLIPS','--reps','5']
from musicfpaugment_amd import _lib
_lib.set_library_path("$l")
exec(open('tools/exp_upconv.py').read())
PY
done; done
