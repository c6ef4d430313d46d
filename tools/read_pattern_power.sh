#!/bin/bash
# Socket power of the two frame-fetch orders (tools/ubench/read_pattern_power.hip), sampled with rocm-smi while each runs.
for p in 0 1 0 1; do
    ./tools/ubench/read_pattern_power $p 5 &
    pid=$!
    sleep 2.0
    for i in 1 2 3 4; do
        rocm-smi --showpower --showclocks --json | python3 -c "import json,sys; c=next(iter(json.load(sys.stdin).values())); print('   pattern $p:', c['Current Socket Graphics Package Power (W)'], 'W  sclk', c['sclk clock speed:'])"
        sleep 0.5
    done
    wait $pid
done
