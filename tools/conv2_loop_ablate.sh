# Development: k_conv2 main-loop ablations (timing only, results are garbage).  Round 3 ran these on the kernel as it stood mid-round
# (compute waves issuing their own LDS-DMA pieces): -DCV2_ABL bits 1 = no DMA in the loop, 2 = no fragment reads, 4 = no MFMAs,
# 16 = row-operand pieces of taps > 0 from the zero page, 32 = weight pieces always from slab 0 (L1 hits) - numbers in LAB_NOTES.md.
# What the current kernel still builds: -DCV2_ABL=64 (row-operand pieces only for tap 0), -DCV2_STORE_G=1/2/4, -DCV2_INV=1,
# -DCV2_NOSYNC=1/2.  Build a variant and point CLIMSIM_HIP_LIB at it:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCV2_STORE_G=4 climsim_amd/csrc/climsim_hip.hip -o climsim_amd/variants/g4.so
# CS_CONV_ABLATE=4 leaves the epilogues out, 8 keeps them without their global stores.
for v in "" "$@"; do
  for ab in 0 8 4; do
    if [ -z "$v" ]; then L=$PWD/climsim_amd/libclimsim_hip.so; else L=$PWD/climsim_amd/variants/$v.so; fi
    [ -f "$L" ] || continue
    echo -n "variant=${v:-base} "; CLIMSIM_HIP_LIB=$L CS_CONV_ABLATE=$ab python tools/cnn_pred_time.py 2>&1 | tail -1
    echo -n "variant=${v:-base} "; CLIMSIM_HIP_LIB=$L CS_CONV_ABLATE=$ab python tools/cnn_train_time.py 512 2>&1 | tail -1
  done
done
