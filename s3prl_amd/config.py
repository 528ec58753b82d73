"""Typed view of the hyper-parameters that select kernel variants on the upstream-encoder path.

Mirrors the subset of the reference config objects that the forward actually reads:

* ``HubertConfig`` / ``HubertPretrainingConfig``  (s3prl/upstream/hubert/hubert_model.py:33-278)
* ``Wav2Vec2Config`` / ``AudioPretrainingConfig`` (s3prl/upstream/wav2vec2/wav2vec2_model.py:2103-2350,3325-3345)
* ``WavLMConfig``                                 (s3prl/upstream/wavlm/WavLM.py:162-245)
* ``MultiresHubertConfig``                        (s3prl/upstream/multires_hubert/hubert_model.py:97-330)

Like ``merge_with_parent`` (s3prl/upstream/utils.py:31-44) unknown keys of a checkpoint's
config dict are dropped, missing ones fall back to the reference defaults.
"""

from __future__ import annotations

import ast
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Tuple

FAMILIES = ("hubert", "wav2vec2", "wavlm", "distiller", "multires_hubert")

# reference default: "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"
DEFAULT_CONV_LAYERS = "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"


def parse_conv_layers(spec) -> List[Tuple[int, int, int]]:
    """The reference ``eval``s this string (hubert_model.py:297, wav2vec2_model.py:2357).

    We only accept the arithmetic-on-literal-lists subset, evaluated without ``eval``.
    """
    if not isinstance(spec, str):
        return [tuple(int(v) for v in t) for t in spec]

    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
            return ev(node.left) + ev(node.right)
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Mult):
            l, r = ev(node.left), ev(node.right)
            return l * r
        if isinstance(node, (ast.List, ast.Tuple)):
            vals = [ev(e) for e in node.elts]
            return vals if isinstance(node, ast.List) else tuple(vals)
        if isinstance(node, ast.Constant) and isinstance(node.value, int):
            return node.value
        raise ValueError(f"unsupported conv_feature_layers expression: {spec!r}")

    layers = ev(ast.parse(spec, mode="eval"))
    out = []
    for t in layers:
        if len(t) != 3:
            raise ValueError("invalid conv definition: " + str(t))
        out.append((int(t[0]), int(t[1]), int(t[2])))
    return out


@dataclass
class EncoderConfig:
    family: str = "hubert"
    conv_layers: List[Tuple[int, int, int]] = field(
        default_factory=lambda: parse_conv_layers(DEFAULT_CONV_LAYERS)
    )
    extractor_mode: str = "default"  # "default" (GroupNorm after conv0) | "layer_norm"
    conv_bias: bool = False
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    layer_norm_first: bool = False
    conv_pos: int = 128
    conv_pos_groups: int = 16
    normalize: bool = False  # task_cfg.normalize: per-utterance waveform layer-norm
    # WavLM only
    relative_position_embedding: bool = False
    num_buckets: int = 320
    max_distance: int = 1280
    gru_rel_pos: bool = False
    # DistilHuBERT (family "distiller", upstream/distiller/model.py:17-80): no LayerNorm between the conv stack and
    # post_extract_proj, and ``pred_heads`` prediction heads (Linear -> GELU -> SplitLinear) on the last layer
    feature_layer_norm: bool = True
    pred_heads: int = 0
    # data2vec-audio (upstream/data2vec, wav2vec2_model.py:2995-3023): pos_conv_depth > 1 replaces the weight-normed
    # positional conv by that many {Conv1d(k = max(3, conv_pos // depth), groups) -> LayerNorm(no affine) -> GELU} blocks
    pos_conv_depth: int = 1
    # eps of the per-utterance waveform normalisation: F.layer_norm's 1e-5 in the fairseq experts (hubert/expert.py:57-58),
    # 1e-7 in Hugging Face's Wav2Vec2FeatureExtractor (hf_hubert / hf_wav2vec2 upstreams)
    wav_norm_eps: float = 1e-5
    # multi-resolution HuBERT (family "multires_hubert", upstream/multires_hubert/hubert_model.py:337-530): a U-net of
    # TransformerEncoders — encoders[i] -> conv adapter (down) ... middle ... conv adapter (up) -> decoders[i] — over the
    # frame rates given by ``label_rate_ratios`` = [up_0, down_0, up_1, down_1, ...].  ``block_layers``: layers of every
    # encoder in execution order (encoders..., middle, decoders...); ``encoder_layers`` is then their sum.
    label_rate_ratios: List[int] = field(default_factory=list)
    block_layers: List[int] = field(default_factory=list)
    conv_adapter_kernel: int = 7          # the reference spells it ``conv_adapator_kernal``
    use_plain_updownsample: bool = False  # ConvDownsampler / ConvUpsampler instead of the two-conv ConvAdapter

    # ---- derived -------------------------------------------------------------------------
    @property
    def conv_dim(self) -> int:
        return self.conv_layers[-1][0]

    @property
    def head_dim(self) -> int:
        return self.encoder_embed_dim // self.encoder_attention_heads

    @property
    def pos_conv_kernel(self) -> int:
        return self.conv_pos if self.pos_conv_depth <= 1 else max(3, self.conv_pos // self.pos_conv_depth)

    @property
    def num_hidden_states(self) -> int:
        """Entries of the default ``hidden_states`` list: layer inputs + encoder output; DistilHuBERT:
        feat_final + every layer output + the prediction heads (distiller/expert.py:43-52)."""
        if self.family == "multires_hubert":  # every block: its layer inputs + its output (multires_hubert/expert.py:49-91)
            return sum(n + 1 for n in self.block_layers)
        return self.encoder_layers + 1 + self.pred_heads

    @property
    def downsample_rate(self) -> int:
        r = 1
        for _, _, s in self.conv_layers:
            r *= s
        return r

    def conv_lengths(self, n: int) -> List[int]:
        """floor((L-k)/s)+1 per layer (wav2vec2_model.py:2615-2616)."""
        out = []
        for _, k, s in self.conv_layers:
            n = (n - k) // s + 1 if n >= k else 0
            out.append(n)
        return out

    def num_frames(self, n: int) -> int:
        return self.conv_lengths(n)[-1]

    def valid_frames(self, length: int, n_max: int) -> int:
        """Number of un-masked frames of an utterance of ``length`` samples in a batch padded to
        ``n_max`` samples.

        hubert / wavlm: ``forward_padding_mask`` (hubert_model.py:454-464, WavLM.py:339-349):
        chunk = n_max // T; frame t is padding iff all samples of its chunk are padding.
        wav2vec2: conv-length formula of ``length`` (wav2vec2_model.py:2652-2669).
        """
        T = self.num_frames(n_max)
        if T <= 0:
            return 0
        if self.family in ("wav2vec2", "distiller"):  # distiller: cal_pad_mask, distiller/model.py:271-285
            return min(T, max(self.num_frames(length), 0))
        chunk = n_max // T
        return min(T, -(-length // chunk))

    # ---- multires-HuBERT geometry ---------------------------------------------------------
    @property
    def rate_pairs(self) -> List[Tuple[int, int]]:
        r = self.label_rate_ratios
        return [(int(r[2 * i]), int(r[2 * i + 1])) for i in range(len(r) // 2)]

    def adapter_frames(self, T: int, up: int, down: int, kind: str) -> int:
        """Output frames of a conv adapter on ``T`` frames (hubert_model.py:1038-1095 ConvAdapter, :1146-1180
        ConvDownsampler, :1232-1266 ConvUpsampler): every stage is cut to min(conv length, skip-connection length)."""
        k = self.conv_adapter_kernel
        n = T
        if kind in ("full", "up"):
            n = min(up * T + k - 1, up * T)  # ConvTranspose1d(padding=0, output_padding=stride-1) vs repeat_interleave
        if kind in ("full", "down"):
            ld = (n + 2 * ((k - 1) // 2) - k) // down + 1
            n2 = min(ld, -(-n // down))
            n = min(n2, -(-(up * T) // down)) if kind == "full" else n2  # highway branch (ConvAdapter only)
        return n

    def multires_plan(self, T0: int):
        """Frame geometry of the U-net for a conv-stack output of ``T0`` frames: a list of blocks
        ``dict(prefix, layers, T, factor, adapter)`` in execution order (adapter = the (kind, up, down, module prefix)
        applied BEFORE the block, None for the first), the per-decoder truncation of ``align_size_sum``
        (hubert_model.py:777-783) as ``T_sum``, and ``T_out`` = the common length the expert cuts every (upsampled)
        state to (multires_hubert/expert.py:26-27,93-101)."""
        import math

        pairs = self.rate_pairs
        R = len(pairs) + 1
        assert len(self.block_layers) == 2 * R - 1
        ds = [self.downsample_rate]
        for u, d in pairs:  # hubert_model.py:512-533
            ds.append(ds[-1] * d // u)
        lcm = 1
        for x in ds:
            lcm = lcm * x // math.gcd(lcm, x)
        upf = [lcm // x for x in ds][::-1]  # (sic) expert.py:44-45
        rev = upf[::-1][1:]
        plain = self.use_plain_updownsample
        blocks, T, enc_T, ad = [], T0, [], None
        for i in range(R - 1):
            blocks.append(dict(prefix=f"encoders.{i}", layers=self.block_layers[i], T=T, factor=upf[i], adapter=ad))
            enc_T.append(T)
            u, d = pairs[i]
            ad = ("down" if plain else "full", u, d, f"downsample_modules.{i}")
            T = self.adapter_frames(T, u, d, ad[0])
        blocks.append(dict(prefix="middle_encoder", layers=self.block_layers[R - 1], T=T, factor=upf[R - 1], adapter=ad))
        res = enc_T[::-1]
        for i in range(R - 1):
            d, u = pairs[i]  # upsample_modules[i] is built from the INVERTED pair i (hubert_model.py:474-507)
            ad = ("up" if plain else "full", u, d, f"upsample_modules.{i}")
            T = self.adapter_frames(T, u, d, ad[0])
            blocks.append(dict(prefix=f"decoders.{i}", layers=self.block_layers[R + i], T=T, factor=rev[i], adapter=ad,
                               T_sum=min(T, res[i])))
            T = min(T, res[i])
        cand = []
        for b in blocks:
            cand.append(b["T"] * b["factor"])
            if b["layers"] > 0:
                cand.append((b["T"] + b["T"] % 2) * b["factor"])  # layer inputs are padded to a multiple of 2
        return blocks, min(cand)

    def num_output_frames(self, n: int) -> int:
        """Frames of every entry of ``hidden_states`` for a batch padded to ``n`` samples."""
        T = self.num_frames(n)
        if self.family != "multires_hubert" or T < 1:
            return T
        return self.multires_plan(T)[1]

    def validate(self) -> None:
        if self.family not in FAMILIES:
            raise ValueError(f"unknown family {self.family!r}")
        if self.family == "multires_hubert":
            pairs = self.rate_pairs
            if not pairs or len(self.label_rate_ratios) % 2:
                raise ValueError("multires_hubert needs label_rate_ratios = [up, down, ...]")
            if len(pairs) > 3:
                raise ValueError("multires_hubert: at most 4 resolutions")
            if len(self.block_layers) != 2 * len(pairs) + 1 or min(self.block_layers) < 1:
                raise ValueError("multires_hubert: block_layers needs one positive entry per encoder / middle / decoder")
            if self.encoder_layers != sum(self.block_layers):
                raise ValueError("multires_hubert: encoder_layers must equal sum(block_layers)")
            k = self.conv_adapter_kernel
            if k < 1 or k % 2 == 0 or k > 15:
                raise ValueError("multires_hubert: conv_adapter_kernel must be odd and <= 15")
            for u, d in pairs:
                if self.use_plain_updownsample and u != 1:
                    raise ValueError("use_plain_updownsample needs label_rate_ratios of the form (1, d)")  # :1130,1214
                for s in (u, d):
                    if s < 1 or s > 4 or (k - 1) % s:
                        raise ValueError("multires_hubert: every rate must divide conv_adapter_kernel - 1 (the transposed "
                                         "conv is run as `rate` interleaved stride-1 convs)")
        if self.extractor_mode not in ("default", "layer_norm"):
            raise ValueError(f"unknown extractor_mode {self.extractor_mode!r}")
        dims = {d for d, _, _ in self.conv_layers}
        if len(dims) != 1:
            raise ValueError("all conv feature layers must have the same width")
        if self.encoder_embed_dim % self.encoder_attention_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        if self.head_dim != 64:
            raise ValueError("the HIP attention kernel is specialised for head_dim == 64")
        if self.encoder_embed_dim % self.conv_pos_groups:
            raise ValueError("embed_dim must be divisible by conv_pos_groups")

    def to_dict(self) -> Dict:
        return asdict(self)


_MODEL_KEYS = (
    "extractor_mode", "conv_bias", "encoder_layers", "encoder_embed_dim", "encoder_ffn_embed_dim",
    "encoder_attention_heads", "layer_norm_first", "conv_pos", "conv_pos_groups",
)
_WAVLM_KEYS = ("relative_position_embedding", "num_buckets", "max_distance", "gru_rel_pos", "normalize")


def config_from_dicts(family: str, model_cfg: Dict, task_cfg: Dict | None = None) -> EncoderConfig:
    """Build an :class:`EncoderConfig` from the dicts stored in a converted checkpoint (§3.4)."""
    cfg = EncoderConfig(family=family)
    for k in _MODEL_KEYS:
        if k in model_cfg and model_cfg[k] is not None:
            setattr(cfg, k, type(getattr(cfg, k))(model_cfg[k]))
    if "conv_feature_layers" in model_cfg:
        cfg.conv_layers = parse_conv_layers(model_cfg["conv_feature_layers"])
    act = model_cfg.get("activation_fn", "gelu")
    act = getattr(act, "name", act)
    if str(act) != "gelu":
        raise ValueError(f"only activation_fn='gelu' is on the hot path, got {act!r}")
    if str(model_cfg.get("layer_type", "transformer")).endswith("conformer"):
        raise ValueError("conformer layers are out of scope (SURVEY §2.1)")
    cfg.pos_conv_depth = int(model_cfg.get("pos_conv_depth", 1) or 1)
    if cfg.pos_conv_depth > 1 and family != "wav2vec2":
        raise ValueError("pos_conv_depth > 1 is the data2vec-audio encoder (wav2vec2 family)")
    if family == "wavlm":
        for k in _WAVLM_KEYS:
            if k in model_cfg:
                setattr(cfg, k, type(getattr(cfg, k))(model_cfg[k]))
    else:
        if task_cfg is not None and "normalize" in task_cfg:
            cfg.normalize = bool(task_cfg["normalize"])
    cfg.validate()
    return cfg


def config_from_multires(model_cfg: Dict, task_cfg: Dict | None = None) -> EncoderConfig:
    """``MultiresHubertConfig`` (upstream/multires_hubert/hubert_model.py:97-330) + ``task_cfg.normalize``."""
    cfg = EncoderConfig(family="multires_hubert")
    for k in _MODEL_KEYS:
        if k in model_cfg and model_cfg[k] is not None:
            setattr(cfg, k, type(getattr(cfg, k))(model_cfg[k]))
    if "conv_feature_layers" in model_cfg:
        cfg.conv_layers = parse_conv_layers(model_cfg["conv_feature_layers"])
    act = model_cfg.get("activation_fn", "gelu")
    if str(getattr(act, "name", act)) != "gelu":
        raise ValueError(f"only activation_fn='gelu' is on the hot path, got {act!r}")
    if str(model_cfg.get("layer_type", "transformer")).endswith("conformer"):
        raise ValueError("conformer layers are out of scope (SURVEY §2.1)")
    ratios = model_cfg.get("label_rate_ratios", [1, 2])
    if ratios in (None, "None"):
        raise ValueError("without ratios, the model is exactly as the Hubert model")  # hubert_model.py:362-364
    cfg.label_rate_ratios = [int(x) for x in ratios]
    n_blocks = len(cfg.label_rate_ratios) // 2 * 2 + 1
    per_block = int(model_cfg.get("encoder_layers", 2))
    over = model_cfg.get("override_encoder_layers", "") or ""
    if over:  # hubert_model.py:377-403,415-424: encoders[i] = o[i], middle = o[len // 2], decoders[i] = o[len - 1 - i]
        o = [int(x) for x in (ast.literal_eval(over) if isinstance(over, str) else over)]
        if len(o) != n_blocks:
            raise ValueError("number of override encoder layers must match the label rate ratios information")
        R = n_blocks // 2 + 1
        cfg.block_layers = o[:R] + [o[len(o) - 1 - i] for i in range(R - 1)]
    else:
        cfg.block_layers = [per_block] * n_blocks
    cfg.encoder_layers = sum(cfg.block_layers)
    cfg.conv_adapter_kernel = int(model_cfg.get("conv_adapator_kernal", 7))
    cfg.use_plain_updownsample = bool(model_cfg.get("use_plain_updownsample", False))
    if task_cfg is not None and "normalize" in task_cfg:
        cfg.normalize = bool(task_cfg["normalize"])
    cfg.validate()
    return cfg


def config_from_distiller(d: Dict) -> EncoderConfig:
    """``DistillerConfig`` (upstream/distiller/model.py:17-80) from ``ckpt["Config"]["distiller"]``."""
    cfg = EncoderConfig(family="distiller", feature_layer_norm=False)
    cfg.extractor_mode = str(d.get("extractor_mode", "default"))
    cfg.conv_layers = parse_conv_layers(d.get("extractor_conv_feature_layers", DEFAULT_CONV_LAYERS))
    cfg.conv_pos = int(d.get("conv_pos", 128))
    cfg.conv_pos_groups = int(d.get("conv_pos_groups", 16))
    cfg.encoder_layers = int(d.get("encoder_layers", 1))
    cfg.encoder_embed_dim = int(d.get("encoder_embed_dim", 768))
    cfg.encoder_ffn_embed_dim = int(d.get("encoder_ffn_embed_dim", 3072))
    cfg.encoder_attention_heads = int(d.get("encoder_attention_heads", 12))
    cfg.layer_norm_first = bool(d.get("layer_norm_first", False))
    if str(d.get("activation_fn", "gelu")) != "gelu":
        raise ValueError("only activation_fn='gelu' is on the hot path")
    if str(d.get("attention_type", "original")) != "original":
        raise ValueError("distiller attention_type must be 'original'")
    task, out = str(d.get("task_emb_type", "expand-last")), str(d.get("out_layer_type", "expand-last"))
    if task != "expand-last" or out != "expand-last":
        raise ValueError("only the DistilHuBERT head layout (task_emb_type = out_layer_type = 'expand-last') is built")
    cfg.pred_heads = int(d.get("n_tasks", 12))
    if int(d.get("final_dim", 768)) != cfg.encoder_embed_dim or int(d.get("out_layer_inter_dim", -1)) > 0:
        raise ValueError("distiller heads must keep the encoder width (final_dim == encoder_embed_dim, no inter dim)")
    if cfg.conv_dim == cfg.encoder_embed_dim:
        raise ValueError("distiller without post_extract_proj (conv width == encoder width) is not built")
    cfg.validate()
    return cfg
