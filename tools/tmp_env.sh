for E in "COMET_ADC_STREAM_TABLES=1" "COMET_ADC_STREAM_TABLES=1 COMET_ADC_LUT_MB=256" "COMET_ADC_STREAM_TABLES=1 COMET_ADC_LUT_MB=128" "COMET_ADC_STREAM_TABLES=1 COMET_ADC_LUT_MB=64"; do
  env $E python bench.py --legs ivfpq_uniform,ivfpq10m --no-cpu-baseline --regions 3 --sustain-s 0.5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
for k, v in d['legs'].items():
    print('$E', k, 'pruned q/s', round(v.get('qps', 0)), 'single', round(v.get('single_stream_qps', 0)), 'every-cand q/s', round(v.get('every_candidate_qps', 0)), 'adc_scan ms', v.get('adc_scan_ms'))
"
done
