import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(d["value"],d["ms_per_step"])
for n in sys.argv[2:]: print(n,d["kernels"].get(n))
