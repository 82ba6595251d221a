#!/bin/bash
# The round's gate: the FULL GPU suite with -x on a fresh lease, exactly as the driver runs it at round
# end, plus smoke() -- and the log kept under profiles/.  Run through gpurun from the repo root AFTER the last kernel / default-flag
# commit of the round:
#   gpurun --timeout 1700 -- "GIT_HEAD=$(git rev-parse HEAD) bash tools/gate.sh r06"
# then `cp gpurun_out/<tag>_gate.txt profiles/<tag>_gate.txt` and commit it.  No kernel or default-flag commit after it.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_gate.txt
cd $R
SHA=$(python -c "from dupl_amd.build import source_digest; print(source_digest())")
{ echo "# gate: python -m pytest tests/ -x -q -m gpu ; python -c 'import __graft_entry__ as g; g.smoke()'"
  echo "# git_head: ${GIT_HEAD:-unknown}"; echo "# csrc_sha256: $SHA"; echo "# date: $(date -u +%FT%TZ)"; } > $OUT
timeout 1500 python -m pytest tests/ -x -q -m gpu >> $OUT 2>&1
echo "# pytest rc: $?" >> $OUT
# the slow cases (minutes of host-side oracle work each; skipped by the plain run above): the 8-image COCO step vs the oracle
echo "# slow: DUPL_RUN_SLOW=1 python -m pytest tests/ -x -q -m 'gpu and slow'" >> $OUT
DUPL_RUN_SLOW=1 timeout 1200 python -m pytest tests/ -x -q -m "gpu and slow" >> $OUT 2>&1
echo "# pytest slow rc: $?" >> $OUT
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("__SMOKE_OK__")' >> $OUT 2>&1
echo "# smoke rc: $?" >> $OUT
tail -8 $OUT
