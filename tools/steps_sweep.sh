for cfg in "20 3" "100 3" "400 100" "20 3"; do
  set -- $cfg
  timeout -k 10 120 python3 bench.py --steps $1 --warmup $2 --no-configs --cpu-spp 0 > gpurun_out/r3_i_steps_$1_$2.json 2> gpurun_out/r3_i_steps.err || exit 1
  python3 -c "
import json;d=json.loads(open('gpurun_out/r3_i_steps_$1_$2.json').read().strip().splitlines()[-1]);print('$1 $2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['resolve_ms'])"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
