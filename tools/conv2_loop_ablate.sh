# Development: k_conv2 main-loop ablations.  Build the variants first (garbage results, timing only):
#   for a in 1 3 4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCV2_ABL=$a climsim_amd/csrc/climsim_hip.hip -o climsim_amd/variants/abl$a.so; done
# CV2_ABL bits: 1 = no DMA in the loop, 2 = no fragment reads, 4 = no MFMAs; CS_CONV_ABLATE=4 leaves the epilogues out.
for v in "" abl64; do
  for ab in 0 4; do
    if [ -z "$v" ]; then L=$PWD/climsim_amd/libclimsim_hip.so; else L=$PWD/climsim_amd/variants/$v.so; fi
    [ -f "$L" ] || continue
    echo -n "variant=${v:-base} "; CLIMSIM_HIP_LIB=$L CS_CONV_ABLATE=$ab python tools/cnn_pred_time.py 2>&1 | tail -1
  done
done
