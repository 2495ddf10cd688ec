#!/bin/bash
# Sample board power and shader clock from sysfs (hwmon power1_average / power1_input, pp_dpm_sclk, freq1_input) every 50 ms while a command runs:
#   bash scripts/power_sample.sh <out.txt> <command ...>
# One line per sample: ms since start, microwatts, current sclk level line(s).  The summary (mean / max over the middle 80 % of the samples) goes to stdout.
out=$1; shift
dev=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
pw=$(ls $dev/hwmon/hwmon*/power1_average $dev/hwmon/hwmon*/power1_input 2>/dev/null | head -1)
fq=$(ls $dev/hwmon/hwmon*/freq1_input 2>/dev/null | head -1)
: > $out
echo "# dev=$dev power=$pw freq=$fq" >> $out
"$@" &
pid=$!
t0=$(date +%s%N)
while kill -0 $pid 2>/dev/null; do
  t=$(( ($(date +%s%N) - t0) / 1000000 ))
  p=$( [ -n "$pw" ] && cat $pw 2>/dev/null )
  f=$( [ -n "$fq" ] && cat $fq 2>/dev/null )
  s=$(grep '\*' $dev/pp_dpm_sclk 2>/dev/null | tr '\n' ' ')
  echo "$t ${p:-NA} ${f:-NA} $s" >> $out
  sleep 0.05
done
wait $pid
python3 - "$out" <<'PY'
import sys
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
rows = rows[len(rows) // 10: len(rows) - len(rows) // 10] or rows
def col(i):
    v = []
    for r in rows:
        try: v.append(float(r[i]))
        except Exception: pass
    return v
p, f = col(1), col(2)
if p: print("power W: mean %.0f max %.0f (n=%d)" % (sum(p) / len(p) / 1e6, max(p) / 1e6, len(p)))
if f: print("freq1_input MHz: mean %.0f min %.0f max %.0f" % (sum(f) / len(f) / 1e6, min(f) / 1e6, max(f) / 1e6))
lv = [" ".join(r[3:]) for r in rows if len(r) > 3]
if lv:
    import collections
    print("sclk levels:", collections.Counter(lv).most_common(4))
PY
