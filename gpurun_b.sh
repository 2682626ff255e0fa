cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "--- order: batch fastest (flags 3)"; DS_ATT_FLAGS=3 python $R/tools/microbench.py attention
echo "--- order: q-block fastest (flags 1)"; DS_ATT_FLAGS=1 python $R/tools/microbench.py attention
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d $R/gpurun_out/pmc_att1 -o a -- python $R/tools/microbench.py attention > $R/gpurun_out/pmc_att1.log 2>&1
DS_ATT_FLAGS=3 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d $R/gpurun_out/pmc_att3 -o a -- python $R/tools/microbench.py attention > $R/gpurun_out/pmc_att3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/pmc_att1b -o a -- python $R/tools/microbench.py attention > $R/gpurun_out/pmc_att1b.log 2>&1
cd $R; python tools/pmc_summary.py gpurun_out/pmc_att1 gpurun_out/pmc_att1b --match attention_fwd; python tools/pmc_summary.py gpurun_out/pmc_att3 --match attention_fwd
