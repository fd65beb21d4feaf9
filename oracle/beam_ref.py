"""TEST INFRASTRUCTURE (see oracle/__init__.py) — CPU restatement of the reference's greedy beam search.

Restates GDR_model/transformers/generation_utils.py:629-921 (`_generate_beam_search`, the
do_sample=False branch) and `BeamHypotheses` (:1052-1099), including the host bookkeeping order,
for the arguments GDR passes (main_models.py:1380-1397): early_stopping=False, min_length=0,
repetition_penalty=1, no n-gram / bad-word bans, eos=1, pad=0.
"""
import torch
import torch.nn.functional as F


class BeamHypotheses:
    """generation_utils.py:1052-1099."""

    def __init__(self, num_beams, length_penalty):
        self.num_beams = num_beams
        self.length_penalty = length_penalty
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / len(hyp) ** self.length_penalty
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                sorted_scores = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[sorted_scores[0][1]]
                self.worst_score = sorted_scores[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        cur_score = best_sum_logprobs / cur_len ** self.length_penalty
        return self.worst_score >= cur_score


def trie_mask(input_ids, decode_tree, vocab_size):
    """The trie constraint of the un-imported earlier generation_utils_previous.py:714-729 (NCI semantics, `--tree 1`):
    -inf on every token that is not a child of the node reached by the row's prefix; a prefix that left the tree may
    only emit EOS.  `decode_tree` is a nested dict {token: subtree} (the reference's Node.children)."""
    mask = torch.full((input_ids.shape[0], vocab_size), float("-inf"))
    for i in range(input_ids.shape[0]):
        cur = decode_tree
        for value in input_ids[i, 1:].tolist():
            if value not in cur:
                nxt = [1]
                break
            cur = cur[value]
        else:
            nxt = list(cur.keys())
        mask[i, nxt] = 0
    return mask


def build_trie(seqs):
    """TreeBuilder.add (main_models.py:135-151) as nested dicts: seqs are token lists with trailing EOS(1), PAD(0) ends."""
    root = {}
    for seq in seqs:
        cur = root
        for tok in seq:
            if tok == 0:
                break
            cur = cur.setdefault(tok, {})
    return root


def beam_search(step_fn, batch_size, num_beams, vocab_size, max_length, length_penalty,
                num_return_sequences=None, eos_token_id=1, pad_token_id=0, start_token_id=0,
                trace=None, decode_tree=None, prefix_trace=None):
    """step_fn(seq int64[B*R, cur_len]) -> next-token logits fp32[B*R, vocab_size] (last position,
    positional mask already applied).  Returns (decoded int64[B*nret, <=max_length], scores list[float]).
    ``trace`` (a list) receives per step (top_scores[B,2R], top_tokens[B,2R]) for golden comparison; ``prefix_trace`` (a
    list) the beams' token prefixes int64[B*R, cur_len] as they stand BEFORE that step — together they let a test explain why a
    hypothesis is absent from the final list (which cut it fell at, and by how much)."""
    R = num_beams
    nret = num_return_sequences or R
    hyps = [BeamHypotheses(R, length_penalty) for _ in range(batch_size)]
    beam_scores = torch.zeros((batch_size, R), dtype=torch.float)
    beam_scores[:, 1:] = -1e9                                             # :663-668
    beam_scores = beam_scores.view(-1)
    done = [False] * batch_size
    input_ids = torch.full((batch_size * R, 1), start_token_id, dtype=torch.long)
    cur_len = 1
    while cur_len < max_length:                                           # :676
        logits = step_fn(input_ids)
        scores = F.log_softmax(logits, dim=-1)                            # :698
        if decode_tree is not None:                                       # generation_utils_previous.py:714-729
            scores = scores + trie_mask(input_ids, decode_tree, vocab_size)
        next_scores = (scores + beam_scores[:, None]).view(batch_size, R * vocab_size)
        next_scores, next_tokens = torch.topk(next_scores, 2 * R, dim=1, largest=True, sorted=True)  # :775
        if trace is not None:
            trace.append((next_scores.clone(), next_tokens.clone()))
        if prefix_trace is not None:
            prefix_trace.append(input_ids.clone())
        next_batch_beam = []
        for b in range(batch_size):                                       # :783
            if done[b]:
                next_batch_beam.extend([(0, pad_token_id, 0)] * R)
                continue
            nxt = []
            for rank, (tok_id, tok_score) in enumerate(zip(next_tokens[b], next_scores[b])):
                beam_id = tok_id // vocab_size
                token_id = tok_id % vocab_size
                eff = b * R + beam_id
                if token_id.item() == eos_token_id:
                    if rank >= R:
                        continue
                    hyps[b].add(input_ids[eff].clone(), tok_score.item())  # :814-817
                else:
                    nxt.append((tok_score, token_id, eff))
                if len(nxt) == R:
                    break
            done[b] = done[b] or hyps[b].is_done(next_scores[b].max().item(), cur_len)   # :827-829
            assert len(nxt) == R, "Beam should always be full"
            next_batch_beam.extend(nxt)
        if all(done):
            break
        beam_scores = beam_scores.new([x[0] for x in next_batch_beam])
        beam_tokens = input_ids.new([x[1] for x in next_batch_beam])
        beam_idx = input_ids.new([x[2] for x in next_batch_beam])
        input_ids = torch.cat([input_ids[beam_idx, :], beam_tokens.unsqueeze(1)], dim=-1)   # :848-849
        cur_len += 1
    for b in range(batch_size):                                           # :863-883
        if done[b]:
            continue
        for beam_id in range(R):
            eff = b * R + beam_id
            hyps[b].add(input_ids[eff], beam_scores[eff].item())
    sent_lengths = input_ids.new(batch_size * nret)
    best, out_scores = [], []
    for i, h in enumerate(hyps):                                          # :895-902
        sorted_hyps = sorted(h.beams, key=lambda x: x[0])
        for j in range(nret):
            score, best_hyp = sorted_hyps.pop()
            sent_lengths[nret * i + j] = len(best_hyp)
            best.append(best_hyp)
            out_scores.append(score)
    sent_max_len = min(sent_lengths.max().item() + 1, max_length)
    decoded = input_ids.new_full((batch_size * nret, sent_max_len), pad_token_id)
    for i, hypo in enumerate(best):                                       # :911-916
        decoded[i, : sent_lengths[i]] = hypo
        if sent_lengths[i] < max_length:
            decoded[i, sent_lengths[i]] = eos_token_id
    return decoded, out_scores


def generate(sd, cfg, input_ids, attention_mask, num_beams, max_length=None, length_penalty=0.8,
             num_return_sequences=None, restricted_head=False, trace=None, decode_tree=None):
    """GenerationMixin.generate as GDR calls it (generation_utils.py:110-527; main_models.py:1380-1397):
    encoder once, expand per beam, full decoder recompute every step (use_cache=False).
    Returns ((decoded, scores), enc_expanded[B*R,L,d])."""
    from . import t5_ref
    B = input_ids.shape[0]
    R = num_beams
    max_length = max_length or cfg.max_output_length
    enc = t5_ref.encoder_forward(sd, cfg, input_ids, attention_mask)
    idx = torch.arange(B).view(-1, 1).repeat(1, R).view(-1)              # :450-461
    enc_x = enc.index_select(0, idx)
    mask_x = attention_mask.index_select(0, idx)

    def step(seq):
        return t5_ref.decode_logits(sd, cfg, seq, enc_x, mask_x, restricted=restricted_head)

    out = beam_search(step, B, R, cfg.decode_vocab_size, max_length, length_penalty,
                      num_return_sequences or R, cfg.eos_token_id, cfg.pad_token_id,
                      cfg.decoder_start_token_id, trace=trace, decode_tree=decode_tree)
    return out, enc_x
