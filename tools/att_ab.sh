# A/B of the attention kernel generations through the C ABI, without Python (tools/att_harness.cpp).  Needs, in the tree:
#     hipcc -O2 -std=c++17 tools/att_harness.cpp -o tools/att_harness -ldl
#     python stable-diffusion-webui-depthmap-script_amd/build_native.py
# Run on the GPU box:  gpurun --timeout 120 -- 'bash tools/att_ab.sh'
# Per shape: generation 2 without / with the GEMV tail blocks, generation 4.
P=stable-diffusion-webui-depthmap-script_amd/libdepthstereo_hip.so
O=gpurun_out/att; mkdir -p $O
run() {  # B H n Np bias tag
  echo "== $6"
  DS_ATT_GEN=2 DS_ATT_TAIL=0 timeout 20 ./tools/att_harness $P $1 $2 $3 $4 $5 20 /tmp/$6_g2.bin 2>&1 | sed 's/^/gen 2 tiled tail  /' | tee $O/$6_g2.txt
  DS_ATT_GEN=2 timeout 20 ./tools/att_harness $P $1 $2 $3 $4 $5 20 /tmp/$6_g2t.bin 2>&1 | sed 's/^/gen 2            /' | tee $O/$6_g2t.txt
  DS_ATT_GEN=4 timeout 20 ./tools/att_harness $P $1 $2 $3 $4 $5 20 /tmp/$6_g4.bin 2>&1 | sed 's/^/gen 4            /' | tee $O/$6_g4.txt
}
run 32 16 1025 1032 1 c3
run 8 16 2443 2448 0 c5
run 8 16 4097 4104 1 c3match
run 32 12 577 584 0 c2b32
run 4 16 1370 1376 0 dav2_518
