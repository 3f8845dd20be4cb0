# round-3 PMC rows of the DB-AIAT prior's kernels (attention, GRU, row LayerNorm, dense-block convolutions):
# gpurun -- 'bash tools/scripts/r03_prof_aia.sh'; summaries are copied to profiles/r03_pmc_aia_* by hand
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c -d $O/pmc_$c -o p --output-format csv -- python3 tools/time_aia.py > /dev/null 2> $O/pmc_$c.log; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_SQ -o p --output-format csv -- python3 tools/time_aia.py > /dev/null 2> $O/pmc_SQ.log
rocprofv3 --kernel-trace --stats -d $O/ks -o p --output-format csv -- python3 tools/time_aia.py > $O/time_aia.txt 2> $O/ks.log
for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/$d/p_counter_collection.csv; done
python tools/summarize_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; tail -30 $O/pmc_traffic.log
python tools/pmc_summary.py $O/pmc_SQ > $O/pmc_sq.txt 2>&1
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 $f | cut -c1-160
