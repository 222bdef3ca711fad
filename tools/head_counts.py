"""Row counts of the SMTP head for the bench batch (M = selected token rows, Lm = label cells) - read back from the device after a step."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
modeling = importlib.import_module("graph-gpt_amd.modeling"); training = importlib.import_module("graph-gpt_amd.training")
synth = importlib.import_module("graph-gpt_amd.synth")
B, S, F, V = 256, 32, 13, 756
cfg = modeling.GraphGPTConfig(hidden_act="gelu", vocab_size=V, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                              max_position_embeddings=1024, causal_attention=False, stacked_feat=F, next_n_token=F, attention_dropout=0.1)
model = modeling.GraphGPTPretrainBase(cfg, seed=0)
model._ensure_engine(B, S)
engine = training.initialize(model, training.OptimConfig(lr=3e-4))
batch = synth.make_pretrain_batch(B=B, S=S, F=F, V=V, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k not in ("lengths", "segments")}
dev["num_tokens"] = int(synth.real_tokens(batch))
for _ in range(3):
    training.batch_training(dev, engine)
    torch.cuda.synchronize()
    print("real tokens", dev["num_tokens"], "head counts (M, Lm)", model._engine.head_counts())
