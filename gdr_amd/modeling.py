"""The reference's Python call surface for the inference hot path, backed by libgdr_hip.so.

Drop-in targets:
  * `GDRModel.generate(...)`           T5ForConditionalGeneration.generate as GDR calls it
                                       (GDR_model/main_models.py:1380-1397, main.py:171-187;
                                        transformers/generation_utils.py:110-527)
  * `GDRModel.get_encoder()(...)`      transformers/modeling_t5.py:1319-1320, generation_utils.py:410-411
  * `QueryEncoder` / `EncoderModel`    GDR_model/main_models.py:79-109 (forward(query_enc=h) -> h[:,0])
  * `DenseModel`                       GDR_model/dense.py:30-54 (encode_query / encode_passage / compute_similarity)
  * `GDRRetriever.validation_step_i`   the two-stage retrieval of main_models.py:1337-1642 (decode -> rerank)
There is no CPU path: every method needs CUDA (ROCm) tensors and raises if the HIP library is missing.
"""
import types

import torch

from . import _ffi, codec, ops
from .config import GDRConfig


class ModelOutput(dict):
    """Minimal stand-in for transformers' BaseModelOutput: attribute + item access (file_utils.py ModelOutput)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def strip_lightning_prefix(state_dict, prefix="model."):
    """A Lightning checkpoint's `state_dict` prefixes T5 keys with `model.` and the doc tower with `encoder.`
    (main_models.py:794,797; main.py:121-126).  Returns the T5 part with the prefix removed."""
    if "state_dict" in state_dict:
        state_dict = state_dict["state_dict"]
    if any(k.startswith(prefix + "shared.") or k.startswith(prefix + "encoder.block") for k in state_dict):
        return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
    return state_dict


class _Encoder:
    """What `model.get_encoder()` returns: callable like T5Stack.forward (modeling_t5.py:685-821)."""

    def __init__(self, handle):
        self.handle = handle

    def __call__(self, input_ids=None, attention_mask=None, return_dict=True, **_ignored):
        h, _ = self.handle.forward(input_ids, attention_mask, want_pooled=False)
        if return_dict:
            return ModelOutput(last_hidden_state=h, past_key_values=None, hidden_states=None, attentions=None)
        return (h,)

    forward = __call__


class GDRModel:
    """T5ForConditionalGeneration of the reference (GDR config: decode_embedding=2, adaptor_efficient),
    inference only.  Construct from a reference-style state_dict (SURVEY Appendix C key names)."""

    def __init__(self, cfg: GDRConfig, state_dict, device="cuda:0", with_decoder=True, trie=None, ragged=False,
                 prefix_trie=None, dtype=torch.float32, graph=False):
        """trie: optional codec.Trie — enables the NCI trie constraint of the reference's earlier
        generation_utils_previous.py:714-729 (the shipped generate() ignores `decode_tree`).
        ragged: generate() skips the PAD rows of the encoder (gdr_t5_encoder_forward_ragged).  Decoded ids, scores and the
        CLS rows are unchanged (cross-attention masks PAD keys, kept rows are bit-identical); only the PAD positions of the
        `last_hidden_state` returned with output_encoder_embedding=True are zero instead of the reference's values there,
        hence opt-in.  get_encoder() always computes every row.
        prefix_trie: optional codec.Trie over the corpus' docids — builds the device prefix table (ops.PrefixTable) at
        load: beams whose prefix is a node of that trie read the query-independent adaptor/head results from it instead of
        recomputing them (modeling_t5.py:1618-1639); other beams are computed as before.  Same logits up to fp32
        summation order.  When `trie` (the constraint) is given too it must be the same trie.
        dtype=torch.bfloat16: BASELINE config C5's precision mode (the reference has none: precision=32) — every linear of
        encoder, decoder, adaptor and head takes bf16 operands with fp32 accumulate; all other arithmetic stays fp32.
        graph: replay the decode (gdr_t5_generate) of a repeated call shape as one captured HIP graph instead of ~1 400
        host launches per call (ops.T5DecoderHandle.generate); same kernels, same results."""
        self.config = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _ffi.GdrError("GDRModel needs a CUDA (ROCm) device; there is no CPU path")
        sd = strip_lightning_prefix(state_dict)
        self.dtype = dtype
        self.enc = ops.T5EncoderHandle(cfg, sd, self.device, dtype=dtype)
        self.dec = ops.T5DecoderHandle(cfg, sd, self.device, dtype=dtype) if with_decoder else None
        self.prefix_table = None
        if prefix_trie is not None:
            if self.dec is None:
                raise _ffi.GdrError("prefix_trie needs the decoder (with_decoder=True)")
            if trie is not None and trie is not prefix_trie:
                raise _ffi.GdrError("trie= and prefix_trie= must be the same codec.Trie object")
            self.prefix_table = ops.PrefixTable(self.dec, prefix_trie, self.device)
        if trie is None:
            self.trie = None
        else:                                     # the constraint shares the table's breadth-first arrays
            self.trie = self.prefix_table.device_trie if self.prefix_table is not None else ops.DeviceTrie(trie, self.device)
        self.graph = bool(graph)
        self.ragged = bool(ragged)
        self.training = False

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def get_encoder(self):
        return _Encoder(self.enc)

    # ------------------------------------------------------------------ generate()
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, max_length=None, num_beams=None, length_penalty=None,
                 num_return_sequences=None, early_stopping=False, use_cache=False, do_sample=False,
                 decode_embedding=None, decode_vocab_size=None, output_scores=False, output_encoder_embedding=False,
                 **model_kwargs):
        """Same kwargs / return shape as the vendored generate(): returns `(output, encoder_outputs | None)` where
        output is `(LongTensor[B*nret, <=max_length], list[float])` with output_scores=True, else the LongTensor
        (generation_utils.py:524-527, 918-921).  Unknown kwargs (decode_tree, decoder_index, cluster_constraint,
        decoder_attention_mask, ...) are accepted and ignored exactly as the reference swallows them
        (modeling_t5.py:1755)."""
        cfg = self.config
        if self.dec is None:
            raise _ffi.GdrError("this GDRModel was built with with_decoder=False")
        assert input_ids is not None, "generate() needs input_ids"
        max_length = max_length if max_length is not None else cfg.max_output_length
        num_beams = num_beams if num_beams is not None else 1
        length_penalty = length_penalty if length_penalty is not None else 1.0
        num_return_sequences = num_return_sequences if num_return_sequences is not None else 1
        # the reference's asserts (generation_utils.py:296-324) for the arguments this path uses
        assert isinstance(max_length, int) and max_length > 0, "`max_length` should be a strictly positive integer."
        assert isinstance(num_beams, int) and num_beams > 0, "`num_beams` should be a strictly positive integer."
        assert length_penalty > 0, "`length_penalty` should be strictly positive."
        assert isinstance(num_return_sequences, int) and num_return_sequences > 0, \
            "`num_return_sequences` should be a strictly positive integer."
        if do_sample:
            raise NotImplementedError("sampling is outside the GDR hot path (gen_method='greedy', main.py:299)")
        if num_beams == 1:
            raise NotImplementedError("GDR always decodes with num_beams = num_return_sequences > 1 (infer.sh:10-15)")
        assert num_return_sequences <= num_beams, "num_return_sequences has to be <= num_beams for greedy beam search"
        if decode_vocab_size is not None:
            assert decode_vocab_size == cfg.decode_vocab_size, "decode_vocab_size does not match the loaded head"
        assert 1 < max_length, "The context has 1 number of tokens, but `max_length` is only %d" % max_length
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        enc_h, ids, lens, scores = self._generate_launch(input_ids, attention_mask, num_beams, max_length, length_penalty,
                                                          num_return_sequences)
        decoded, score_list = ops.finish_generate_output(ids, lens, scores, max_length)
        output = (decoded, score_list) if output_scores else decoded
        if output_encoder_embedding:
            # the reference hands back the states already expanded per beam (generation_utils.py:459-461);
            # callers stride them with [::num_beams] (main_models.py:1466)
            expanded = enc_h.repeat_interleave(num_beams, dim=0)
            return output, ModelOutput(last_hidden_state=expanded)
        return output, None

    def _generate_launch(self, input_ids, attention_mask, num_beams, max_length, length_penalty, num_return_sequences):
        """The device part of generate(): encoder + beam decode enqueued on the current stream, nothing read back.
        Returns (encoder states [B,L,d], ids, lens, scores) — device tensors that ops.finish_generate_output turns into
        what generate() returns."""
        input_ids, attention_mask = input_ids.to(self.device), attention_mask.to(self.device)
        enc_h, _ = self.enc.forward(input_ids, attention_mask, want_pooled=False, ragged=self.ragged)
        ids, lens, scores = self.dec.generate(enc_h, attention_mask, num_beams, max_length, length_penalty,
                                              num_return_sequences, trie=self.trie, prefix_table=self.prefix_table,
                                              graph=self.graph)
        return enc_h, ids, lens, scores


class EncoderModel:
    """main_models.py:62-109.  `encoder(query_enc=hidden) -> hidden[:, 0]` (CLS pool; `output` is None in the reference);
    `encoder(passage={'input_ids','attention_mask'[, 'token_type_ids']}) -> pooler_output` of the DPR/BERT doc tower
    (modeling_dpr.py:146-191) when built with `bert=` (an ops.BertEncoderHandle)."""

    def __init__(self, output=None, bert=None, ragged=False):
        self.output, self.bert, self.ragged = output, bert, ragged

    @staticmethod
    def from_state_dict(bcfg, state_dict, device, prefix="ctx_encoder.bert_model.", dtype=torch.float32, ragged=None, split=False):
        """state_dict with the reference's doc-tower keys; a Lightning checkpoint prefixes them `encoder.model.`.
        ragged: PAD positions of a padded batch are not computed (bit-identical pooled output in fp32; r06).  dtype=torch.bfloat16:
        the bf16 precision mode (bf16 linear operands, fp32 accumulate), which exists in the ragged form only."""
        sd = state_dict.get("state_dict", state_dict)
        lp = "encoder.model."
        if any(k.startswith(lp) for k in sd):
            sd = {k[len(lp):]: v for k, v in sd.items() if k.startswith(lp)}
        ragged = (dtype == torch.bfloat16 or bool(split)) if ragged is None else ragged
        return EncoderModel(bert=ops.BertEncoderHandle(bcfg, sd, device, prefix, dtype=dtype, split=split), ragged=ragged)

    def __call__(self, passage=None, query_enc=None):
        if passage is not None:
            if self.bert is None:
                raise _ffi.GdrError("EncoderModel was built without doc-tower weights (use EncoderModel.from_state_dict)")
            p = {k: v.view(-1, v.size(-1)) for k, v in passage.items()}              # main_models.py:81-82
            _, pooled = self.bert.forward(p["input_ids"], p.get("attention_mask"), p.get("token_type_ids"),
                                          want_hidden=False, ragged=self.ragged)
            return pooled
        return self.encode_query(query_enc)

    def encode_passage(self, psg):
        return None if psg is None else self(passage=psg)

    forward = __call__

    def encode_query(self, qry_hidden):
        if qry_hidden is None:
            return None
        return qry_hidden[:, 0] if self.output is None else self.output(q=qry_hidden)


class DensePooler:
    """dense.py:10-27: rep = Linear(h[:,0]) (+ L2 normalise)."""

    def __init__(self, weight_q, bias_q, weight_p=None, bias_p=None, normalize=False):
        self.wq, self.bq = weight_q, bias_q
        self.wp, self.bp = (weight_p if weight_p is not None else weight_q), (bias_p if bias_p is not None else bias_q)
        self.normalize = normalize

    def __call__(self, q=None, p=None):
        if q is not None:
            rep = ops.linear(q[:, 0].contiguous(), self.wq, epilogue=_ffi.EPI_BIAS, bias=self.bq)
        elif p is not None:
            rep = ops.linear(p[:, 0].contiguous(), self.wp, epilogue=_ffi.EPI_BIAS, bias=self.bp)
        else:
            raise ValueError
        if self.normalize:
            rep = ops.l2_normalize(rep)
        return rep


class DenseModel:
    """dense.py:30-54 bi-encoder contract over two callables returning `.last_hidden_state`."""

    def __init__(self, lm_q, lm_p=None, pooler=None):
        self.lm_q, self.lm_p, self.pooler = lm_q, (lm_p if lm_p is not None else lm_q), pooler
        self._ws = None

    def encode_passage(self, psg):
        if psg is None:
            return None
        h = self.lm_p(**psg, return_dict=True).last_hidden_state
        return self.pooler(p=h) if self.pooler is not None else h[:, 0]

    def encode_query(self, qry):
        if qry is None:
            return None
        h = self.lm_q(**qry, return_dict=True).last_hidden_state
        return self.pooler(q=h) if self.pooler is not None else h[:, 0]

    def compute_similarity(self, q_reps, p_reps):
        """scores = q_reps @ p_reps.T (dense.py:53-54) — materialises [B,N]; prefer search() for top-k."""
        return ops.linear(q_reps.contiguous(), p_reps.contiguous())

    def __call__(self, query=None, passage=None):
        """EncoderModel.forward, eval branch (encoder.py:77-113): EncoderOutput(q_reps, p_reps[, scores], loss=None)."""
        q_reps, p_reps = self.encode_query(query), self.encode_passage(passage)
        if q_reps is None or p_reps is None:
            return ModelOutput(q_reps=q_reps, p_reps=p_reps, loss=None, scores=None)
        return ModelOutput(loss=None, scores=self.compute_similarity(q_reps, p_reps), q_reps=q_reps, p_reps=p_reps)

    forward = __call__

    def search(self, q_reps, p_reps, k):
        """compute_similarity + topk(k) fused (never writes the score matrix). Returns (values, int64 indices).
        p_reps: the passage embeddings [N,d] (fp32, or bf16 = the C5 precision mode), or an ops.PrefilteredCorpus built from them once
        (the same fp32 top-k through the bf16 pre-filter, gdr_sim_topk_prefilter)."""
        if self._ws is None:
            self._ws = ops.Workspace(q_reps.device)
        v, i = ops.sim_topk(q_reps.contiguous(), p_reps, k, workspace=self._ws, exact_on_overflow=True)
        return v, i.to(torch.int64)


class LazyRows:
    """A list that is built on first use.  GDRRetriever's step output names clusters and docs as STRINGS, as the reference's
    does (main_models.py:1398,1629-1633) — at 512 queries x 7 alphas x 10 docs that is 40 000 Python strings per step, several
    milliseconds of host time that a caller who only consumes `rerank_values` / the id tensors (or a benchmark loop) never
    needs.  Behaves like the list it stands for: len, indexing, iteration, == with a list or another LazyRows."""

    def __init__(self, n, make):
        self._n, self._make, self._val = n, make, None

    def _get(self):
        if self._val is None:
            self._val = self._make()
            self._make = None
        return self._val

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())

    def __eq__(self, other):
        return self._get() == (other._get() if isinstance(other, LazyRows) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __repr__(self):
        return repr(self._get())


class GDRRetriever:
    """Two-stage GDR retrieval = `T5FineTuner.validation_step_i` (main_models.py:1337-1642):
    beam-decode cluster ids -> id_mapping lookup -> tanh(q·d) over the candidates -> + alpha*softmax(beam scores)
    per cluster -> top-k, for every alpha in score_rate."""

    def __init__(self, model: GDRModel, doc_embed, cluster_index: codec.ClusterIndex, args, doc_tower=None,
                 doc_tokens=None, device_candidates=True, sharded=None):
        """doc_embed: fp32 (or, in the C5 precision mode, bf16) [N, d] resident on the GPU (the reference's `self.doc_embed`).
        doc_tower + doc_tokens=(input_ids int64[N,Lp], attention_mask) enable the stage-2 re-encode path of
        main_models.py:1445-1455 (`epoch > train_encoder_epoch`): candidate docs are embedded on the fly by the
        BERT/DPR tower instead of being looked up (tokenisation itself is out of scope: tokens come pre-computed).
        sharded: a dist.ShardedIndex — BASELINE config C5's layout of this path (SURVEY §8e "for GDR mode, whole clusters"):
        the corpus is row-sharded over the ranks of one node (`doc_embed` may then be None: the rank's rows are
        `sharded.D`), every rank encodes and beam-decodes ITS OWN queries (generate() is data-parallel, as the reference's
        per-GPU launch is: Data_process/NQ_dataset/bert/bert_NQ.sh:5-12), and stage 2 is `sharded.rerank_own` — one
        all-gather of the queries + candidate blocks, per-shard scoring, one all-to-all, merge: the lists are bit-identical
        to the unsharded rerank.  Every rank must call validation_step_i the same number of times with the same batch size
        (the collectives are fixed-size); `cluster_index` is the index of the WHOLE corpus on every rank."""
        self.model, self.args, self.index = model, args, cluster_index
        self.doc_embed = doc_embed if doc_embed is not None or sharded is None else sharded.D
        self.sharded = sharded
        if sharded is not None and not device_candidates:
            raise _ffi.GdrError("GDRRetriever(sharded=...) exchanges the device candidate blocks: device_candidates must stay on")
        self.encoder = doc_tower if doc_tower is not None else EncoderModel()
        self.doc_tokens = doc_tokens
        # device_candidates: decoded rows -> clusters -> candidate CSR on the GPU (gdr_cluster_candidates); False keeps the
        # host form (decode_token strings + ClusterIndex.candidates) — same lists, one D2H / H2D round trip more per step
        self.device_candidates = bool(device_candidates)

    def _reencode(self, cand_ids, chunk=1024):
        """Embeds the candidate docs with the doc tower (main_models.py:1445-1455).  cand_ids int32 [total] on device."""
        tok, msk = self.doc_tokens
        out = []
        for lo in range(0, cand_ids.numel(), chunk):
            sel = cand_ids[lo:lo + chunk].long()
            out.append(self.encoder(passage={"input_ids": tok[sel], "attention_mask": msk[sel]}))
        return torch.cat(out, dim=0)

    @torch.no_grad()
    def validation_step_i(self, batch, i=-1, reencode=False):
        """batch: {"source_ids", "source_mask"} (+ optionally what the reference's dataset adds for the metric rows:
        "texts" list[str] (the reference decodes them from source_ids with the T5 tokenizer — out of scope here),
        "gt" list[str] gold cluster ids, "rank" list[int], "oldid" list[str] gold doc ids).
        Returns the reference's step output {"inf_result_batch", "inf_result_batch_prob", "inf_index_batch"}
        (main_models.py:1640-1641; rows are filled when "texts" is given) plus the raw pieces: "clusters" [B][R] decoded
        cluster strings, "doc_ids" [B][A][R] doc ids as strings, "rerank_values" fp32[B,A,R]."""
        return self._step_finish(self._step_launch(batch), reencode=reencode)

    @torch.no_grad()
    def validation_steps(self, batches, depth=2, reencode=False):
        """The same steps for a sequence of batches with up to `depth` of them in flight, each on a HIP stream of its own:
        while the host decodes batch k's docids, looks up its candidates and formats its rows (the part of
        validation_step_i that needs the beam output on the host, main_models.py:1398-1462), the GPU already runs batch
        k+1's encoder and beam decode; per-stream scratch (ops.Workspace) keeps the calls apart.  Yields the step outputs
        in order — identical to calling validation_step_i batch by batch."""
        if depth <= 1:
            for b in batches:
                yield self.validation_step_i(b, reencode=reencode)
            return
        if not hasattr(self, "_streams") or len(self._streams) < depth:
            self._streams = [torch.cuda.Stream(device=self.model.device) for _ in range(depth)]
        cur = torch.cuda.current_stream(self.model.device)
        pending, k = [], 0
        for b in batches:
            st = self._streams[k % depth]
            st.wait_stream(cur)                              # inputs prepared on the caller's stream
            with torch.cuda.stream(st):
                pending.append((st, self._step_launch(b)))
            k += 1
            if len(pending) == depth:
                st0, state = pending.pop(0)
                with torch.cuda.stream(st0):
                    out = self._step_finish(state, reencode=reencode)
                yield out
        for st0, state in pending:
            with torch.cuda.stream(st0):
                out = self._step_finish(state, reencode=reencode)
            yield out

    def _step_launch(self, batch):
        a = self.args
        R = a.num_return_sequences
        mask = batch["source_mask"] if batch.get("source_mask") is not None else torch.ones_like(batch["source_ids"])
        enc_h, ids, lens, scores = self.model._generate_launch(batch["source_ids"], mask, R, a.max_output_length,
                                                               a.length_penalty, R)
        for t in (enc_h, ids, lens, scores):                 # produced on this stream, possibly consumed after a switch
            if t.is_cuda:
                t.record_stream(torch.cuda.current_stream(t.device))
        return {"batch": batch, "enc_h": enc_h, "ids": ids, "lens": lens, "scores": scores}

    def _device_index(self):
        """The cluster index on the GPU (ops.DeviceClusterIndex), built on first use; None when the id scheme has no
        separator (--kary 0): the host lookup below then serves, as in round 2."""
        if not hasattr(self, "_dci"):
            a = self.args
            self._dci = None
            if getattr(a, "kary", 30) and self.device_candidates:
                self._dci = ops.DeviceClusterIndex(self.index, self.model.device, getattr(a, "output_vocab_size", a.kary),
                                                   position=getattr(a, "position", 1), kary=a.kary)
        return self._dci

    def _step_finish(self, state, reencode=False):
        a = self.args
        R = a.num_return_sequences
        batch = state["batch"]
        query_embeds = self.encoder(query_enc=state["enc_h"]).contiguous()      # CLS rows (main_models.py:1466)
        B = query_embeds.shape[0]
        alphas, func = list(a.score_rate), getattr(a, "loss_func", "tanh")
        dci = self._device_index()
        stride = 0
        if dci is not None:
            # decode_token -> id_mapping -> candidate lists -> rerank, all enqueued before anything is read back
            # (main_models.py:1398,1441-1443,1574-1637): the host only formats strings afterwards
            _cl, offs, dev_ids, stride = dci.candidates(state["ids"], B, R)
            # the sharded stage 2 takes its bound from the GATHERED offsets (dist.ShardedIndex): a rank-local decision here could
            # raise on one rank while its peers already sit in the fixed-size all-gather
            max_cand = ops.block_max_cand(offs, R, stride) if self.sharded is None else 0
            beam_scores = state["scores"].to(torch.float32).view(B, R)          # fp64 -> fp32 as torch.tensor(list) rounds
        outs = scores = None
        if dci is None:
            outs, scores = ops.finish_generate_output(state["ids"], state["lens"], state["scores"], a.max_output_length)
            dec = codec.dec_2d(codec.decode_token(a, outs.cpu().numpy()), R)
            offs, ids, max_cand = self.index.candidates(dec)
            offs = offs.to(query_embeds.device)
            beam_scores = torch.tensor(scores, dtype=torch.float32, device=query_embeds.device).view(B, R)
            dev_ids = ids.to(query_embeds.device)
        if self.sharded is not None:
            if dci is None or reencode:
                raise _ffi.GdrError("the sharded two-stage path needs the device cluster index (--kary > 0) and has no "
                                    "re-encode form (the doc tower's tokens are not sharded)")
            vals, idx = self.sharded.rerank_own(query_embeds, offs, dev_ids, beam_scores, alphas, R, func=func)
        elif reencode:
            if self.doc_tokens is None or getattr(self.encoder, "bert", None) is None:
                raise _ffi.GdrError("re-encode needs doc_tower= and doc_tokens=")
            if stride:                                   # block layout: the live ids of every query, query-major
                cnt = offs[:, R].long()
                live_mask = torch.arange(stride, device=cnt.device)[None, :] < cnt[:, None]
                live = dev_ids[live_mask]
                start = torch.cumsum(cnt, 0) - cnt
                local = (start[:, None] + torch.arange(stride, device=cnt.device)[None, :]).to(torch.int32)
            else:
                live = dev_ids[:max(int(offs[-1].item()), 1)]
                local = torch.arange(live.numel(), dtype=torch.int32, device=dev_ids.device)
            if live.numel() == 0:
                live = dev_ids.reshape(-1)[:1]
            cand_embeds = self._reencode(live)                                      # [total, d], candidate order
            vals, pos = ops.rerank_topk(query_embeds, cand_embeds, offs, local, beam_scores, alphas, R, func=func,
                                        max_cand=max_cand, cand_stride=stride)
            idx = torch.where(pos >= 0, live[pos.clamp(min=0).long()], pos)         # candidate position -> doc id
        else:
            vals, idx = ops.rerank_topk(query_embeds, self.doc_embed, offs, dev_ids, beam_scores, alphas, R, func=func,
                                        max_cand=max_cand, cand_stride=stride)
        if outs is None:                                                            # first read-back of the step
            outs, scores = ops.finish_generate_output(state["ids"], state["lens"], state["scores"], a.max_output_length)
        outs_h = outs.cpu().numpy()
        idx_np = idx.cpu().numpy()                                                  # [B, A, R] doc ids (-1 = padding)
        _ffi.check_device_fault("validation_step_i")             # everything of this step has been read back
        A = len(a.score_rate)
        if dci is not None:
            dec = LazyRows(B, lambda: codec.dec_2d(codec.decode_token(a, outs_h), R))
        doc_ids = LazyRows(B, lambda: [[list(map(str, row)) for row in q] for q in idx_np.tolist()])
        inf_result, inf_index = [], []
        texts = batch.get("texts")
        if texts is not None:
            gts = batch.get("gt", [""] * B)
            ranks = batch.get("rank", [1] * B)
            old = batch.get("oldid", [""] * B)
            idx_h = idx_np.tolist()
            for b in range(B):                                   # main_models.py:1421-1432 and :1626-1637
                inf_result.append([texts[b], ",".join(dec[b]), gts[b], int(ranks[b])])
                inf_index.append([[[texts[b], ",".join(map(str, idx_h[b][ai])), old[b]]] for ai in range(A)])
        return {"inf_result_batch": inf_result, "inf_result_batch_prob": scores, "inf_index_batch": inf_index,
                "clusters": dec, "doc_ids": doc_ids, "rerank_values": vals, "doc_id_tensor": idx}
