import json, sys
r = json.load(open(sys.argv[1]))
print({k: r.get(k) for k in ("value", "host_to_host_Mq_s", "stream_Mq_s", "estimator_Mq_s")})
