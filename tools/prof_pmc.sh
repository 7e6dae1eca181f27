#!/usr/bin/env bash
# Hardware counters per kernel on the GPU box (one rocprofv3 --pmc pass per counter group, no
# trace domains besides the kernel trace):
#   tools/prof_pmc.sh <name> "<group1 counters>" "<group2 counters>" ... -- <command ...>
# writes gpurun_out/<name>_pmc.txt (mean counter value per launch, per kernel name).
set -uo pipefail
name=$1; shift
groups=()
while [[ $# -gt 0 && "$1" != "--" ]]; do groups+=("$1"); shift; done
shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$repo/gpurun_out"
export TMPDIR=/tmp
rm -rf /tmp/pmc_$name; mkdir -p /tmp/pmc_$name
i=0
for g in "${groups[@]}"; do
  ( cd "$repo" && rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc_$name/g$i -o run -- "$@" ) \
    > /tmp/pmc_$name/g$i.log 2>&1
  i=$((i+1))
done
python "$repo/tools/pmc_summary.py" /tmp/pmc_$name > "$repo/gpurun_out/${name}_pmc.txt" 2>&1
cat "$repo/gpurun_out/${name}_pmc.txt"
