import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
r = bench.other_config("C4_encoder_h768", torch.device("cuda:0"))
print("h768", r["ms_per_step"], r["roofline"])
r = bench.other_config("C4_encoder", torch.device("cuda:0"))
print("h128", r["ms_per_step"])
