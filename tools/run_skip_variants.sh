# product-code skip variants of conv_wd16_kernel's 64-channel form (compile-time MFPA_SKIP_BITS), all in one call on one box
for v in "" _skip32 _skip64 _skip128 _skip256 _skip480 _skip8 ""; do
  echo "== libmfpa$v.so"
  python tools/exp_c64.py --lib musicfpaugment_amd/libmfpa$v.so 2>/dev/null | grep -v "amdgpu\|DBG"
done
