cd $GRAFT_REPO_ROOT
run() { echo -n "cfg $1: "; C8_ONLY=1 DKT_C8_CFG=$1 python tools/c8_loop_check.py 2>&1 | grep "ms/pair"; }
run 1,2,3,4,2,2,3
run 1,3,3,4,2,2,3
run 1,2,4,4,2,2,3
run 1,2,3,3,2,2,3
run 1,2,3,4,1,2,3
run 1,2,3,4,3,2,3
run 1,2,3,4,2,3,3
run 1,2,3,4,2,2,4
run 6,2,3,4,2,2,3
run 2,2,3,4,2,2,3
