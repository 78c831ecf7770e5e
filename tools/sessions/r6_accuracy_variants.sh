cd $GRAFT_REPO_ROOT
for lib in base noprescale nor2tiny nosqrt6 noln2 norcp3 nofreeze; do
  if [ $lib = base ]; then unset GPV_LIB; else export GPV_LIB=$GRAFT_REPO_ROOT/gpvecchia_amd/libgpvecchia_hip_$lib.so; fi
  echo "=== $lib"
  python tools/accuracy_posterior_probe.py post:9298 vl:142 vl:102 2>&1 | grep -v amdgpu.ids
done
