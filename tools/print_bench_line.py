import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], d["value"], d["ms_per_step"], d["device_ms_per_step"], d["single_stream"]["value"])
