"""Generates tests/golden/qwen2_hf_tiny.npz: a tiny random-init Qwen2 (the HuggingFace `transformers` implementation, i.e.
the upstream definition of the architecture whose checkpoints xllm/models/llm/qwen2.h loads) run in fp32 on CPU --
prompt, greedy continuation and the logits of every generated token -- together with its weights in the reference's
fused layout (qkv = [q_proj; k_proj; v_proj], gate_up = [gate_proj; up_proj]: Qwen2AttentionImpl::load_state_dict,
layers/common/qwen2_attention.cpp:195-199, DenseMLPImpl::load_state_dict, dense_mlp.cpp:118-121).

The fixture pins the COMPOSITION of the oracle model (oracle/model.py: operator order, GQA head mapping, neox RoPE, fused
projections, residual handling) on an implementation that is not ours; the per-operator cast points are pinned elsewhere.
Run here (needs transformers, which does not have to exist where the tests run):  python tests/golden/make_hf_golden.py
"""
import os

import numpy as np
import torch
from transformers import Qwen2Config, Qwen2ForCausalLM

CFG = dict(vocab_size=320, hidden_size=64, intermediate_size=160, num_hidden_layers=3, num_attention_heads=4,
           num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-6, rope_theta=1e4)
PROMPT_LEN, N_NEW = 37, 8


def main():
    torch.manual_seed(20260924)
    cfg = Qwen2Config(tie_word_embeddings=False, attn_implementation="eager", **CFG)
    m = Qwen2ForCausalLM(cfg).eval().float()
    with torch.no_grad():   # default init leaves norms at 1 and biases at 0: make every parameter matter
        for n, p in m.named_parameters():
            if n.endswith("layernorm.weight") or n.endswith("norm.weight"):
                p.copy_(torch.rand_like(p) + 0.5)
            elif n.endswith(".bias"):
                p.copy_(torch.randn_like(p) * 0.1)
            else:
                p.copy_(torch.randn_like(p) / np.sqrt(p.shape[-1]))
        m.model.embed_tokens.weight.copy_(torch.randn_like(m.model.embed_tokens.weight))
    prompt = torch.randint(0, CFG["vocab_size"], (1, PROMPT_LEN))
    toks, logits = [], []
    with torch.no_grad():
        out = m(prompt, use_cache=True)
        past, lg = out.past_key_values, out.logits[0, -1]
        for _ in range(N_NEW):
            t = int(lg.argmax())
            toks.append(t)
            logits.append(lg.clone())
            out = m(torch.tensor([[t]]), past_key_values=past, use_cache=True)
            past, lg = out.past_key_values, out.logits[0, -1]
    sd = m.state_dict()
    arrs = dict(prompt=prompt[0].numpy().astype(np.int32), tokens=np.asarray(toks, np.int32),
                logits=torch.stack(logits).numpy(), embed=sd["model.embed_tokens.weight"].numpy(),
                norm_w=sd["model.norm.weight"].numpy(), lm_head=sd["lm_head.weight"].numpy())
    for i in range(CFG["num_hidden_layers"]):
        p = f"model.layers.{i}."
        cat = lambda names, suf: torch.cat([sd[p + n + suf] for n in names]).numpy()
        arrs[f"l{i}.qkv_w"] = cat(["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"], ".weight")
        arrs[f"l{i}.qkv_b"] = cat(["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"], ".bias")
        arrs[f"l{i}.o_w"] = sd[p + "self_attn.o_proj.weight"].numpy()
        arrs[f"l{i}.gate_up_w"] = cat(["mlp.gate_proj", "mlp.up_proj"], ".weight")
        arrs[f"l{i}.down_w"] = sd[p + "mlp.down_proj.weight"].numpy()
        arrs[f"l{i}.input_norm_w"] = sd[p + "input_layernorm.weight"].numpy()
        arrs[f"l{i}.post_norm_w"] = sd[p + "post_attention_layernorm.weight"].numpy()
    for k, v in CFG.items():
        arrs["cfg." + k] = np.asarray(v)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qwen2_hf_tiny.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes; tokens", toks)


if __name__ == "__main__":
    main()
