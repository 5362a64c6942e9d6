#!/usr/bin/env python3
"""which rows of the dual-micro-batch step differ from the single-batch step, and does a half-batch single-stream step agree?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from xllm_amd import layers  # noqa: E402
from xllm_amd.attention import AttentionMetadata, KVCache  # noqa: E402

DEV = "cuda"
B = int(os.environ.get("DIAG_B", "512"))
args = layers.ModelArgs(1024, 3, 16, 4, 128, 2048, 4096, 1e-6, 1e6, 4096)
ctx, bs = 300, 128
model = layers.Qwen2Model(args, "int8", torch.bfloat16, DEV, seed=11, n_layers=3)
md, n_blocks = bench.build_metadata(B, ctx, bs, torch.device(DEV), seed=2)
g = torch.Generator(device=DEV).manual_seed(7)
caches0 = [(torch.randn(n_blocks, bs, 4, 128, device=DEV, generator=g).bfloat16(),
            torch.randn(n_blocks, bs, 4, 128, device=DEV, generator=g).bfloat16()) for _ in model.layers]
fresh = lambda: [KVCache(k.clone(), v.clone()) for k, v in caches0]
tokens = torch.randint(0, args.vocab_size, (B,), device=DEV, generator=g)
positions = torch.full((B,), ctx - 1, dtype=torch.int64, device=DEV)
ref = model.forward(tokens, positions, md, fresh()).clone()
ref2 = model.forward(tokens, positions, md, fresh()).clone()
print("single batch deterministic:", torch.equal(ref, ref2))


def half_md(b, e):
    return AttentionMetadata(q_cu_seq_lens=torch.arange(e - b + 1, dtype=torch.int32, device=DEV), kv_cu_seq_lens=None,
                             kv_seq_lens=md.kv_seq_lens[b:e].contiguous(), slot_mapping=md.slot_mapping[b:e].contiguous(),
                             block_table=md.block_table[b:e].contiguous(), max_query_len=1, max_seq_len=md.max_seq_len)


def rows_diff(a, b):
    d = (a.float() - b.float()).abs().amax(-1)
    bad = (d > 0).nonzero().flatten().tolist()
    return len(bad), bad[:8], bad[-4:], float(d.max())


h = B // 2
c = fresh()
seq = torch.cat([model.forward(tokens[b:e], positions[b:e], half_md(b, e), c).clone() for b, e in ((0, h), (h, B))])
print("two half batches, one stream, vs whole batch: differing rows", rows_diff(seq, ref))
dual = layers.DualBatchDecoder(model, md, B)
out = dual.forward(tokens, positions, fresh())
torch.cuda.synchronize()
print("dual vs whole batch: differing rows", rows_diff(out, ref))
print("dual vs two sequential halves:", rows_diff(out, seq))
# layer by layer on the whole batch vs halves: first operator that differs
from xllm_amd import ops  # noqa: E402
x_full = torch.nn.functional.embedding(tokens, model.embed)
L = model.layers[0]
cw, ch = fresh(), fresh()
qw, rw = L.pre_attention(x_full, None, positions, md, cw[0], model.cos_sin)
qh = [L.pre_attention(x_full[b:e], None, positions[b:e], half_md(b, e), ch[0], model.cos_sin) for b, e in ((0, h), (h, B))]
print("layer 0 pre_attention q:", rows_diff(torch.cat([t[0] for t in qh]).view(B, -1), qw.view(B, -1)))
print("layer 0 kv cache after the write:", torch.equal(cw[0].k_cache, ch[0].k_cache), torch.equal(cw[0].v_cache, ch[0].v_cache))
ow = L.attention_kernel(qw, md, cw[0])
oh = [L.attention_kernel(qh[i][0], half_md(b, e), ch[0]) for i, (b, e) in enumerate(((0, h), (h, B)))]
cat = lambda ts: torch.cat([t if not isinstance(t, tuple) else t[0] for t in ts])
print("layer 0 attention out:", rows_diff(cat(oh).view(B, -1).float(), (ow[0] if isinstance(ow, tuple) else ow).view(B, -1).float()))
