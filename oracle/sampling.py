"""TEST INFRASTRUCTURE ONLY (never imported by xllm_amd/): CPU restatement of the reference's logits processors, op for op.

Follows xllm/core/framework/sampling/logits_utils.cpp: apply_frequency_presence_penalties :24-36, apply_repetition_penalties :38-52,
apply_temperatures :54-64, apply_top_k_top_p_torch_impl :66-90 (both top_k and top_p given), apply_top_k_top_p :92-155 (its
"one of them" branch :121-153 is what a CUDA / DCU build runs). Every function is the reference's torch expression restated with the
same operators in the same order (gather / sub_ / scatter_, sort / masked_fill_ / softmax / cumsum / scatter_), on CPU tensors.

Parity pinned: the reference holds no golden vector for these functions (tests/core/framework/sampling/ has sampling_params_test
and rejection_sampler_test only); the restatement is the same torch calls, and tests/test_oracle_sampling.py pins its semantics on
hand-computed cases (which ranks survive for a given p, the k <= 0 rule -- "no limit" under both rules since round 5 --, the
"at least one" rule of the both-given branch).
One deliberate choice: torch.sort(descending=True) is called with stable=True so that ties have a defined order (by column index);
the reference's unstable sort leaves it unspecified.
"""
import torch


def apply_frequency_presence_penalties(logits, unique_token_ids, unique_token_counts, frequency_penalties, presence_penalties):
    score = logits.gather(1, unique_token_ids)
    score.sub_(unique_token_counts * frequency_penalties.unsqueeze(1))
    score.sub_((unique_token_counts > 0) * presence_penalties.unsqueeze(1))
    logits.scatter_(1, unique_token_ids, score)


def apply_repetition_penalties(logits, unique_token_ids, penalties):
    p = penalties.unsqueeze(1)
    score = logits.gather(1, unique_token_ids)
    logits.scatter_(1, unique_token_ids, torch.where(score < 0, score * p, score / p).to(logits.dtype))


def apply_temperatures(logits, temperatures):
    t = temperatures.unsqueeze(1)
    t = torch.where(t == 0, torch.tensor(1.0), t)
    logits.div_(t)


def apply_top_k_top_p_torch_impl(logits, top_k, top_p, nonpositive_k_is_unlimited=False):
    """logits_utils.cpp:61-84 as written (k clamped to [1, vocab]); with nonpositive_k_is_unlimited the k <= 0 rows keep every
    column, as the reference's NPU both-given branch (:109-116) and its one-of-them branch (:126-133) do"""
    vocab = logits.size(-1)
    srt, idx = logits.sort(dim=-1, descending=True, stable=True)
    k = top_k.unsqueeze(-1)
    if nonpositive_k_is_unlimited:
        k = torch.where(k <= 0, torch.tensor(vocab, dtype=k.dtype), k)
    k = k.clamp(1, vocab).to(torch.long)
    k_mask = torch.arange(vocab).expand_as(srt) >= k
    srt.masked_fill_(k_mask, float("-inf"))
    p = top_p.unsqueeze(-1)
    probs = srt.softmax(-1)
    cum = probs.cumsum(-1)
    p_mask = cum > p
    p_mask[..., 0] = False
    srt.masked_fill_(p_mask, float("-inf"))
    logits.scatter_(-1, idx, srt)


def apply_top_k_top_p(logits, temperatures, top_k, top_p):
    """returns the processed logits (the reference reassigns `logits` in its else branch)"""
    if temperatures is not None:
        apply_temperatures(logits, temperatures)
    if top_k is None and top_p is None:
        return logits
    if top_k is not None and top_p is not None:
        # what this backend applies: torch_impl's inclusive prefix + "at least one", with k <= 0 = no limit (the reference's
        # default top_k = -1 must not turn into top-1 in a mixed batch; on CUDA / DCU the reference's both-given branch is EMPTY,
        # :108-119, so there is no behaviour to match beyond the branches that exist)
        apply_top_k_top_p_torch_impl(logits, top_k, top_p, nonpositive_k_is_unlimited=True)
        return logits
    srt, idx = logits.sort(dim=-1, descending=True, stable=True)
    if top_k is not None:
        k = top_k.unsqueeze(1)
        k = torch.where(k <= 0, torch.tensor(torch.iinfo(torch.int64).max), k)
        mask = torch.arange(logits.size(-1)).expand_as(srt) >= k
        srt.masked_fill_(mask, float("-inf"))
    if top_p is not None:
        p = top_p.unsqueeze(1)
        probs = srt.softmax(-1).to(torch.float32)
        probs_sum = probs.cumsum(-1)
        mask = (probs_sum - probs) > p
        srt.masked_fill_(mask, float("-inf"))
    return torch.empty_like(srt).scatter_(-1, idx, srt)
