cd $GRAFT_REPO_ROOT
bash tools/r04_ab.sh build_ab/libazg_base.so alpha-zero-general_amd/libazg_hip.so
