python bench.py > gpurun_out/r03_h_bench.json 2> gpurun_out/r03_h_bench.err
bash tools/prof_stats.sh r03_h python bench.py --steps 100 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
bash tools/prof_stats.sh r03_h_b100 python bench.py --batch 100 --steps 300 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
bash tools/prof_stats.sh r03_h_cfg3 python bench.py --likelihood "zero-inflated negative binomial" --latent 100 --steps 40 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
bash tools/prof_stats.sh r03_h_cfg4 python bench.py --model gmvae --latent 100 --batch 512 --steps 60 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
bash tools/prof_stats.sh r03_h_eval python tools/bench_eval.py > /dev/null 2>&1
bash tools/prof_pmc.sh r03_h "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" -- python bench.py --no-cpu-baseline --no-other-workloads --steps 10 --warmup 2 > /dev/null 2>&1
grep -E "scvae" gpurun_out/r03_h_kernel_stats.txt | head -8
grep -E "decoder_head3" gpurun_out/r03_h_pmc.txt | head -3
grep "evaluation step" gpurun_out/r03_h_eval_cmd.log
