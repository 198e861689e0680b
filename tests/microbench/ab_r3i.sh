V=$GRAFT_REPO_ROOT/tests/microbench/build
cd $GRAFT_REPO_ROOT
for lib in "" "$V/libvar_kx2.so" "$V/libvar_kx1.so"; do
  echo "== lib=[$lib]"
  ARPEGGIA_AMD_LIB=$lib timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/kx_tmp.json 2> gpurun_out/kx_tmp.err || tail -3 gpurun_out/kx_tmp.err
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/kx_tmp.json"))
def row(name, s):
    k = s["roofline"]["kernels_ms"]
    print("%-8s ms/step %.4f  pipeline %.1f us  %s" % (name, s["ms_per_step"], s["roofline"]["pipeline_ms"]*1e3, {a: round(b*1000,1) for a,b in k.items()}))
row("s2", d)
for n in ("s1", "s2_1e5", "s1_1e5", "batch5k"): row(n, d[n])
for n in ("1ubq", "6bft"): print(n, round(d["files"][n]["us_per_call_on_stream"],1), {a: round(b,1) for a,b in d["files"][n]["kernels_us"].items()})
PY
done
