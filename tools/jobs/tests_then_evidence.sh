# GPU suite, then the round's evidence (collect_evidence.sh)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/evidence
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/evidence/tests.log 2>&1 || { tail -60 gpurun_out/evidence/tests.log; exit 1; }
tail -2 gpurun_out/evidence/tests.log
bash tools/jobs/collect_evidence.sh
