#!/usr/bin/env python3
"""ISA-level experiment for profiles/r02_pk_fma_op_sel.md (run where hipcc is; the code objects then travel to the GPU box).
Compiles csrc/demucs.hip with -DMFPA_HEAD_PACKED_FMA to assembly, and writes code objects of c1_glu_kernel that differ from
what hipcc emitted ONLY as named:
  dm_v0                 as emitted (28 x v_pk_fma_f32 ... op_sel:[0,1,0])
  v6_no_opsel010        each of those replaced by  v_mov_b32 v236, <high half>  +  v_pk_fma_f32 ... v[236:237] ... op_sel_hi:[1,0,1]
  v9_vgpr240_only       control: only the VGPR count of the kernel descriptor raised to 240, as v6 needs
  v7_nop7_both          s_nop 7 before and after each op_sel:[0,1,0] instruction
  v2_nop_before, v4_nop_after_vmov   s_nop 3 before each such instruction / after each  v_mov_b32 v4, <odd sample>
usage: make_hsaco_variants.py   ->  tools/probes/hsaco/*.hsaco ; then on the GPU:  run_hsaco_head.py tools/probes/hsaco/*.hsaco"""
import os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "probes", "hsaco"); os.makedirs(OUT, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
K = "_ZN12_GLOBAL__N_113c1_glu_kernelEPKfiiS1_S1_S1_S1_Pfi"
s_path = os.path.join(OUT, "dm.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DMFPA_HEAD_PACKED_FMA=1", "-S", "--cuda-device-only",
                       os.path.join(ROOT, "musicfpaugment_amd", "csrc", "demucs.hip"), "-o", s_path])
src = open(s_path).read()
a = src.index(K + ":"); b = src.index("s_endpgm", a)
k = src.index(".amdhsa_kernel " + K); ke = src.index(".end_amdhsa_kernel", k)
body, desc = src[a:b], src[k:ke]
nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
desc240 = re.sub(r"\.amdhsa_next_free_vgpr \d+", ".amdhsa_next_free_vgpr 240", re.sub(r"\.amdhsa_accum_offset \d+", ".amdhsa_accum_offset 240", desc))
assert nv <= 236, nv
form = re.compile(r"^\tv_pk_fma_f32 (v\[\d+:\d+\]), (v\[\d+:\d+\]), v\[(\d+):(\d+)\], (v\[\d+:\d+\]) op_sel:\[0,1,0\]$", re.M)
any010 = re.compile(r"^(\tv_pk_fma_f32 .*op_sel:\[0,1,0\].*)$", re.M)
vmov = re.compile(r"^(\tv_mov_b32_e32 v4, v\d+)$", re.M)
variants = {
    "dm_v0": (body, desc),
    "v6_no_opsel010": (form.sub(lambda m: f"\tv_mov_b32_e32 v236, v{m.group(4)}\n\tv_pk_fma_f32 {m.group(1)}, {m.group(2)}, v[236:237], {m.group(5)} op_sel_hi:[1,0,1]", body), desc240),
    "v9_vgpr240_only": (body, desc240),
    "v7_nop7_both": (any010.sub(r"\ts_nop 7\n\1\n\ts_nop 7", body), desc),
    "v2_nop_before": (any010.sub(r"\ts_nop 3\n\1", body), desc),
    "v4_nop_after_vmov": (vmov.sub(r"\1\n\ts_nop 3", body), desc),
}
print(len(form.findall(body)), "instructions of the form in the kernel")
for name, (bd, ds) in variants.items():
    p = os.path.join(OUT, name)
    open(p + ".s", "w").write(src[:a] + bd + src[b:k] + ds + src[ke:])
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", p + ".s", "-o", p + ".o"])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", p + ".o", "-o", p + ".hsaco"])
    os.remove(p + ".s"); os.remove(p + ".o")
os.remove(s_path)
print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".hsaco")))
