import sys, json
d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["achieved"])
