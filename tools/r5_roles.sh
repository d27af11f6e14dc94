#!/bin/bash
# round 5: cfg2 step by placement of the engine's four streams on the process's hardware queues (development library)
cd "$(dirname "$0")/.."
for rep in 1 2; do for k in ${ROLES:-2301 2310 2354 2367 2304 2315 2356 2300 2311 6701 6745}; do echo "roles $k: $(ORYON_ENGINE_ROLES=$k python tools/engine_timeline.py 40 2>&1 | grep 'ms/step')"; done; done
