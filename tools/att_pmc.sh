# Hardware counters of the attention kernel generations on the harness (tools/att_harness.cpp), one counter group per pass:
#     gpurun --timeout 300 -- 'bash tools/att_pmc.sh'        -> gpurun_out/att_pmc/<shape>_<gen>.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/stable-diffusion-webui-depthmap-script_amd/libdepthstereo_hip.so
O=$R/gpurun_out/att_pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
one() {  # tag gen B H n Np bias
  tag=$1; gen=$2; shift 2
  i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    (cd /tmp && DS_ATT_GEN=$gen timeout 120 rocprofv3 --pmc $c -d $O/p_${tag}_${gen}_$i -o a -- $R/tools/att_harness $P "$@" 3 /tmp/x.bin > $O/${tag}_${gen}_$i.log 2>&1)
  done
  python $R/tools/pmc_summary.py $O/p_${tag}_${gen}_* --match attention_fwd > $O/${tag}_g${gen}.json 2>&1
  rm -rf $O/p_${tag}_${gen}_*
  python - <<PY
import json
d=json.load(open("$O/${tag}_g${gen}.json"))
for k,v in d.items():
    if not isinstance(v, dict): continue
    print("$tag gen $gen", k[:40])
    wc=v.get("SQ_WAVE_CYCLES",0)
    for c in sorted(v):
        print("   %-28s %14.0f  %s" % (c, v[c], ("%.1f %% of wave cycles" % (100*v[c]/wc)) if wc and c.startswith("SQ_") and "INSTS" not in c else ""))
PY
}
#one c3 2 32 16 1025 1032 1
one c3 4 32 16 1025 1032 1
one c5 4 8 16 2443 2448 0
