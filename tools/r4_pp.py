import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), d["timing"]["stream_busy_ms_per_step"])
