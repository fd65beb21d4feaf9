"""Model hyper-parameters of the GDR T5 (the subset the inference hot path reads).

Mirrors the kwargs the reference forwards into ``T5Config`` (reference:
GDR_model/main_models.py:748-780) and the defaults of
GDR_model/transformers/configuration_t5.py:77-115.
"""
from dataclasses import dataclass, asdict


@dataclass
class GDRConfig:
    vocab_size: int = 32128
    d_model: int = 768
    d_kv: int = 64
    d_ff: int = 3072
    num_heads: int = 12
    num_layers: int = 12              # encoder blocks
    num_decoder_layers: int = 6
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128   # hard-coded in modeling_t5.py:243
    layer_norm_epsilon: float = 1e-6
    pad_token_id: int = 0
    eos_token_id: int = 1
    decoder_start_token_id: int = 0
    # docid head (main_models.py:741-742, modeling_t5.py:1241-1244)
    output_vocab_size: int = 30       # V (= --kary)
    max_output_length: int = 10
    decode_vocab_size: int = 302      # V * max_output_length + 2
    adaptor_layer_num: int = 4
    adaptor_nhead: int = 8            # nn.TransformerDecoderLayer(nhead=8)
    adaptor_ff: int = 2048            # torch default dim_feedforward
    adaptor_ln_eps: float = 1e-5      # torch default layer_norm_eps

    @property
    def inner_dim(self) -> int:
        return self.num_heads * self.d_kv

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def base(**over):
        return GDRConfig(**over)

    @staticmethod
    def tiny(**over):
        """Small shape used by parity tests (every kernel path, seconds on CPU)."""
        kw = dict(vocab_size=128, d_model=64, d_kv=16, d_ff=128, num_heads=4,
                  num_layers=2, num_decoder_layers=2, output_vocab_size=6,
                  max_output_length=5, decode_vocab_size=6 * 5 + 2,
                  adaptor_layer_num=2, adaptor_nhead=8, adaptor_ff=96)
        kw.update(over)
        return GDRConfig(**kw)

    @staticmethod
    def from_args(args):
        """From a reference-style argparse namespace (main.py:422-442 sizes by --model_info).  Raises SystemExit on a
        model variant the kernels do not implement (`unsupported_variant`) — never a silently different model."""
        why = unsupported_variant(args)
        if why:
            raise SystemExit("gdr_amd: " + why)
        V = args.output_vocab_size
        L = args.max_output_length
        return GDRConfig(d_model=args.d_model, d_kv=getattr(args, "d_kv", 64), d_ff=args.d_ff,
                         num_heads=args.num_heads, num_layers=args.num_layers,
                         num_decoder_layers=args.num_decoder_layers,
                         output_vocab_size=V, max_output_length=L,
                         decode_vocab_size=V * L + 2,
                         adaptor_layer_num=args.adaptor_layer_num)


# Reference model variants selected by flags that main.py forwards into T5Config (main_models.py:748-780) and that change
# the arithmetic of the decode branch (modeling_t5.py:1578-1640) or of the validation step (main_models.py:1350-1397).
# The kernels implement exactly ONE of them — the shipped infer.sh / train.sh setting.  Anything else must fail loudly:
# (flag, the value the kernels implement, what the reference would compute instead)
_ONLY = [
    ("adaptor_decode", 1, "without the adaptor the head is the plain lm_head (modeling_t5.py:1640-1644)"),
    ("adaptor_efficient", 1, "adaptor_efficient=0 runs the per-position adaptor of modeling_t5.py:1578-1600"),
    ("decode_embedding", 2, "decode_embedding 0/1 decodes over the T5 vocabulary / a shared docid table "
                            "(main_models.py:1357-1374, modeling_t5.py:1266-1272)"),
    ("hierarchic_decode", 0, "hierarchic_decode=1 shares one V-column head over all positions (main_models.py:741,554)"),
    ("multiple_decoder", 0, "multiple_decoder=1 decodes with decoder_num separate decoders (main_models.py:1376-1379)"),
    ("denoising", 0, "denoising=1 changes the decoder inputs (main_models.py:932, modeling_t5.py)"),
    ("tie_decode_embedding", 1, "tie_decode_embedding=0 unties lm_head from decode_embeddings (modeling_t5.py:1266-1272)"),
    ("softmax", 0, "softmax=1 builds a decoder-less classifier (main_models.py:750,822; validation asserts it off, :1350)"),
    ("gen_method", "greedy", "only gen_method='greedy' (= the beam search of main.py:169-187) is implemented; the "
                             "validation step asserts it (main_models.py:1350)"),
    ("position", 1, "position=0 uses a 12-column position-free docid head (main_models.py:740-745)"),
]


def unsupported_variant(args):
    """None when `args` selects the model variant the HIP path implements, else one sentence naming the first flag that
    does not (reference: main_models.py:748-780 forwards these into T5Config; modeling_t5.py:1578-1640 branches on them)."""
    for name, only, what in _ONLY:
        if hasattr(args, name) and getattr(args, name) != only:
            return (f"--{name} {getattr(args, name)!r} selects a reference model variant this build does not implement "
                    f"(implemented: {only!r}; {what})")
    if getattr(args, "model_info", "base") in ("3b", "11b"):
        return (f"--model_info {args.model_info}: the reference sets no sizes for 3b / 11b (main.py:422-442 covers small / base / "
                "large only, so args.d_kv is undefined at main_models.py:757); refusing to run a base-sized model under that name")
    return None
