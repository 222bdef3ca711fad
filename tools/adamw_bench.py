"""AdamW step micro-benchmark (HIP events): the base model's 122 M parameters through gget_adamw_step (gradient norm + update)."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec_mod = importlib.import_module("graph-gpt_amd.spec"); eng = importlib.import_module("graph-gpt_amd.engine")
spec = spec_mod.spec_from_size("base", vocab_size=756, stacked_feat=13, next_n_token=13)
e = eng.Engine(spec, max_tokens=256, max_batch=8)
e.grad_bf16.normal_(0, 1e-3)
for _ in range(3): e.adamw_step(1e-4, 0.9, 0.95, 1e-8, 0.1, 1.0, 1.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): e.adamw_step(1e-4, 0.9, 0.95, 1e-8, 0.1, 1.0, 1.0)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"adamw step (norm + update) {us:.1f} us for {e.n_params / 1e6:.1f} M elements = {28.0 * e.n_params / us / 1e6:.2f} TB/s of state traffic")
