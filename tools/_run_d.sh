python bench.py > gpurun_out/r03_f_bench.json 2> gpurun_out/r03_f_bench.err
tail -c 3000 gpurun_out/r03_f_bench.json
bash tools/prof_stats.sh r03_f python bench.py --steps 100 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
bash tools/prof_stats.sh r03_f_b100 python bench.py --batch 100 --steps 300 --no-cpu-baseline --no-other-workloads > /dev/null 2>&1
grep scvae gpurun_out/r03_f_kernel_stats.txt | head -30
