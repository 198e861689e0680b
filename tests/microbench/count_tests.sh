cd $GRAFT_REPO_ROOT; export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_count.so
for w in s2 s1; do
  timeout -k 10 200 python bench.py --steps 1 --warmup 0 --profile-steps 0 --no-cpu-baseline --no-extras --workload $w 2>&1 >/dev/null | grep "prefilter tests" | tr "|" "\n" | grep prefilter | tail -1 | sed "s/^/$w 1e6: /"
done
