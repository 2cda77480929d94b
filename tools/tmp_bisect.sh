for E in "A=1" "COMET_COPY_STREAM=1" "A=1" "COMET_COPY_STREAM=1"; do
env $E python bench.py --legs flat,hybrid --no-cpu-baseline --regions 5 --sustain-s 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$E', 'flat', round(d['value']), 'sustained', round(d['sustained_qps']), 'single', round(d['single_stream_qps']), 'ivf', round(d['legs']['hybrid']['ivf_nprobe32']['qps']), round(d['legs']['hybrid']['ivf_nprobe1']['qps']))
"
done
