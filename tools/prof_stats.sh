#!/usr/bin/env bash
# Per-kernel time summary of a command on the GPU box:
#   tools/prof_stats.sh <output name under gpurun_out/> <command ...>
# (rocprofv3 --kernel-trace --stats; run from /tmp as the guide prescribes)
set -uo pipefail
name=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
( cd "$repo" && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- "$@" ) \
  > "$repo/gpurun_out/${name}_cmd.log" 2>&1
db=$(find /tmp/prof_$name -name '*.db' | head -1)
python "$repo/tools/rocprof_summary.py" "$db" > "$repo/gpurun_out/${name}_kernel_stats.txt" 2>&1
head -45 "$repo/gpurun_out/${name}_kernel_stats.txt"
