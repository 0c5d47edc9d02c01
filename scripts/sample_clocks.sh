# Ad-hoc: samples the core / memory clocks and socket power (rocm-smi) while bench.py runs; prints the mean over the samples taken
# during the timed steps.  usage (on a GPU box): bash scripts/sample_clocks.sh [env assignments for bench.py ...]
out=$(mktemp)
( env "$@" python bench.py --legs none --steps 200 --warmup 3 > $out.json 2>/dev/null ) &
pid=$!
sleep 14
n=0
while kill -0 $pid 2>/dev/null && [ $n -lt 60 ]; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | sed 's/GPU\[0\]\s*: //' >> $out
  n=$((n+1))
done
wait $pid
python - "$out" "$@" <<'PY'
import sys, re, json, collections
vals = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.search(r"(sclk|mclk|fclk).*\((\d+)Mhz\)", l)
    if m: vals[m.group(1)].append(int(m.group(2)))
    m = re.search(r"Power.*?:\s*([\d.]+)", l)
    if m: vals["power_w"].append(float(m.group(1)))
d = json.load(open(sys.argv[1] + ".json"))
r = d["roofline"]
print(" ".join(sys.argv[2:]) or "(default)", "| step %.1f ms, k_fused3 %.1f us |" % (d["ms_per_step"], r["kernel_ms"] * 1e3),
      ", ".join("%s mean %.0f min %.0f max %.0f (n=%d)" % (k, sum(v) / len(v), min(v), max(v), len(v)) for k, v in sorted(vals.items())))
PY
