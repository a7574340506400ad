cd $GRAFT_REPO_ROOT
run() { echo -n "cfg $1: "; C8_ONLY=1 DKT_C8_CFG=$1 python tools/c8_loop_check.py 2>&1 | grep "ms/pair"; }
# zr08,q08,zr16,q16,head,enc,c2
run 1,2,4,4,2,3,3
run 1,6,4,4,2,3,3
run 1,2,4,4,2,3,3
run 1,6,4,4,2,3,3
run 1,2,4,6,2,3,3
run 1,2,3,3,2,3,3
run 1,2,4,4,2,6,3
run 1,2,4,4,2,3,4
