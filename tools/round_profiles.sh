#!/usr/bin/env bash
# Per-round profiles of the final build, on the GPU box (writes gpurun_out/<prefix>_*; the summaries
# are copied to profiles/ by hand afterwards).
#   tools/round_profiles.sh                      prefix r04e: the library's default arithmetic, the
#                                              exact nine-term heads (what bench.py's headline runs)
#   tools/round_profiles.sh r04 --head-arith bf16x6   the opt-in six-term heads (profiles/r04_*)
#   QUICK=1: headline only.
#   kernel stats (rocprofv3 --kernel-trace --stats) of the benchmark step for the headline NB VAE,
#   the Poisson VAE, cfg3 (ZINB VAE, latent 100), cfg4 (NB GMVAE K = 20), cfg5 (ZINB GMVAE K = 20,
#   27 998 genes), the headline model at 100 cells; hardware counters (three separate --pmc
#   passes: FETCH_SIZE | WRITE_SIZE | SQ) for the same five training workloads.
cd "$(dirname "$0")/.."
P=${1:-r04e}; shift || true
B="python bench.py --no-cpu-baseline --no-other-workloads $*"
SQ="SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
run() {   # name, bench arguments
  local name=$1; shift
  tools/prof_stats.sh ${P}_${name} $B --steps 60 "$@" > /dev/null 2>&1
  tools/prof_pmc.sh ${P}_${name} "FETCH_SIZE" "WRITE_SIZE" "$SQ" -- $B --steps 10 --warmup 2 "$@" > /dev/null 2>&1
  echo "== $name"; head -8 gpurun_out/${P}_${name}_kernel_stats.txt | cut -c1-150
  grep -E "decoder_head|^kernel" gpurun_out/${P}_${name}_pmc.txt | cut -c1-220
}
run headline
[[ -n "${QUICK:-}" ]] && exit 0
run poisson --likelihood poisson
run cfg3 --likelihood "zero-inflated negative binomial" --latent 100
run cfg4 --model gmvae --latent 100 --batch 512
run cfg5 --model gmvae --latent 100 --batch 512 --likelihood "zero-inflated negative binomial" --features 27998 --cells 16384
tools/prof_stats.sh ${P}_b100 $B --steps 300 --batch 100 > /dev/null 2>&1
echo "== b100"; head -14 gpurun_out/${P}_b100_kernel_stats.txt | cut -c1-150
