"""Scratch diagnostics run on the GPU box (not part of the test-suite)."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _util import load_case, tb, rel_l2
from oracle import gget_oracle as O
eng_mod = importlib.import_module("graph-gpt_amd.engine")
L = importlib.import_module("graph-gpt_amd._lib")

def case(name):
    z, spec, state, batch = load_case(name)
    b = tb(batch)
    B, S = b["input_ids"].shape[:2]
    e = eng_mod.Engine(spec, B * S, B)
    e.load_state_dict(state)
    kind = "pt" if name.startswith("pt") else "ft"
    if kind == "pt":
        loss = e.forward_pretrain(b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt"))
        fn = lambda p: O.pretrain_forward(spec, p, b["input_ids"], b["attention_mask"], b["labels"], b.get("wgt")); lk = "head1_loss"
        logits = None
    else:
        reg = spec.num_labels == 1
        loss, logits, _ = e.forward_task(b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"], None,
                                         L.PROBLEM_REGRESSION_L1 if reg else L.PROBLEM_SINGLE_LABEL)
        fn = lambda p: O.task_forward(spec, p, b["input_ids"], b["attention_mask"], b["position_ids"], b["task_labels"],
                                      problem_type="regression" if reg else "single_label_classification", loss_type="l1" if reg else None); lk = "task_loss"
    e.backward(); torch.cuda.synchronize()
    st_bf = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in state.items()}
    p = O.to_params(st_bf, torch.float32)
    out, grads = O.loss_and_grads(fn, p, lk)
    print(f"== {name}: loss hip {loss.item():.6f} oracle(bf16 weights, fp32 math) {out[lk].item():.6f} ref fp32 {float(z['loss']):.6f} ref bf16 {float(z['loss_bf16']):.6f}")
    if logits is not None:
        print("   task logits hip", logits.cpu().numpy().ravel(), "oracle", out["task_logits"].detach().numpy().ravel(), "labels", b["task_labels"].numpy().ravel())
    g = e.grads()
    gmax = max(float(v.norm()) for v in grads.values())
    for k in state:
        w = grads[k].numpy(); got = g[k].float().cpu().numpy()
        print(f"   {k:50s} |ref| {np.linalg.norm(w):.3e} |hip| {np.linalg.norm(got):.3e} rel {rel_l2(got, w):.4f}")

for n in sys.argv[1:]:
    case(n)
