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
        """From a reference-style argparse namespace (main.py:422-442 sizes by --model_info)."""
        V = args.output_vocab_size
        L = args.max_output_length
        return GDRConfig(d_model=args.d_model, d_kv=getattr(args, "d_kv", 64), d_ff=args.d_ff,
                         num_heads=args.num_heads, num_layers=args.num_layers,
                         num_decoder_layers=args.num_decoder_layers,
                         output_vocab_size=V, max_output_length=L,
                         decode_vocab_size=V * L + 2,
                         adaptor_layer_num=args.adaptor_layer_num)
