cd $GRAFT_REPO_ROOT
for v in ps_c32 ps_c16 ps_c32 ps_c16; do
echo "== $v"
IMPDAR_HIP_LIB=$PWD/build/diag/lib_$v.so python3 profiles/tools/ps_quick.py 8192 | tail -1
done
