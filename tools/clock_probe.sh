# Developer tool (GPU box): shader clock / power while the headline bench loops (is the VALU-bound kernel clock-throttled?)
python bench.py --steps 60000 --warmup 100 --no-cpu-baseline > /tmp/b.json 2>/dev/null &
BP=$!
sleep 6
for k in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | head -8
  echo ---
  sleep 0.5
done
wait $BP
cut -c1-200 /tmp/b.json
rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -3
