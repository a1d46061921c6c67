#!/bin/bash
# GPU box helper: block-encode kernel time per frame against the number of frames per launch (frames per step = 8 launches).
for n in "$@"; do
python bench.py --quick --no-cpu-baseline --frames-per-step $n > gpurun_out/fpl_$n.json 2>/dev/null
python -c "
import json
j=json.load(open('gpurun_out/fpl_$n.json'))
print('frames/step $n (%.1f per launch): value %.1f  kernel ms/frame %.5f  launches %s'%($n/float(j['config']['launches_of_dominant_kernel_per_step']), j['value']/1e3, j['roofline']['kernel_ms_per_frame'], j['config']['launches_of_dominant_kernel_per_step']))"
done
