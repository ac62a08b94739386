for l in "$@"; do echo "=== $l"; FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/$l.so timeout -k 10 300 python3 tools/stamp_pp.py 2>&1 | grep -A1 "512->512\|bn=False" | grep "group 0\|->" ; done
