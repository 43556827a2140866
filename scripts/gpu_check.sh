# GPU box: a bounded test run (every leg under its own timeout -- a hang must not eat the call's limit)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TAG=$1; shift
timeout 900 python -m pytest "$@" -q 2>&1 | grep -E "^E  |passed|failed|FAILED|rror" | tail -25 > gpurun_out/${TAG}.log
