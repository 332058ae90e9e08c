"""EVA02CLIP -- mirror of ape/modeling/text/clip_wrapper_eva02.py:8-150: the `model_language` of the APE configs
(`L(EVA02CLIP)(clip_model="EVA02-CLIP-bigE-14-plus", cache_dir=..., dtype="float16")`), with the text tower on the HIP
kernels (ape_amd/modeling/text/eva02_clip.py) and a per-vocabulary prompt cache.

`forward_text(text_list, cache=False)` returns the reference's dict: `last_hidden_state_eot` [K, embed] (what the vision
model consumes, deformable_detr_segm_vl.py:258-260), `end_token_idx`, `attention_mask` and `last_hidden_state`.  The last
one is computed for the positions up to the longest text of the call (see eva02_clip.py) and zero beyond -- pass
`all_positions=True` at construction for the reference's full 77-position tensor.  Texts are processed in chunks of
`max_batch_size` (:94-111).  dtype: "float32" runs the exact-math validation kernels; "float16" / "bfloat16" run the
production kernels (bf16 storage, fp32 accumulate) and return fp32 features."""
import torch
import torch.nn as nn

from .eva02_clip import TEXT_CONFIGS, CustomCLIPText
from .tokenizer import get_tokenizer


class _TextFeatures(dict):
    """the forward_text dict of the reference; "last_hidden_state" materialises on first access"""
    _lazy_full = None

    def __missing__(self, key):
        if key == "last_hidden_state" and self._lazy_full is not None:
            self[key] = self._lazy_full()
            self._lazy_full = None
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key == "last_hidden_state" and self._lazy_full is not None)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def _materialise(self):
        if self._lazy_full is not None:
            self["last_hidden_state"]           # __missing__ computes and stores it

    # every whole-dict view materialises first, so iteration / items() / values() / len() / copy() / dict(ret) / pickling all see
    # the four keys of the reference's dict; only [] / get / in on the OTHER keys stay lazy
    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        self._materialise()
        return dict.__len__(self)

    def copy(self):
        self._materialise()
        return dict(self)

    def __reduce__(self):
        self._materialise()
        return (dict, (dict(dict.items(self)),))


class EVA02CLIP(nn.Module):
    def __init__(self, clip_model="EVA02-CLIP-B-16", cache_dir=None, dtype="float32", max_batch_size=2560, freeze=True,
                 text_cfg=None, embed_dim=None, tokenizer=None, all_positions=False):
        super().__init__()
        if text_cfg is None:
            if clip_model not in TEXT_CONFIGS:
                raise ValueError(f"ape_amd EVA02CLIP: unknown clip_model {clip_model!r} (known: {sorted(TEXT_CONFIGS)})")
            cfg = dict(TEXT_CONFIGS[clip_model])
            embed_dim = cfg.pop("embed_dim")
            text_cfg = cfg
        self.net = CustomCLIPText(embed_dim, text_cfg)
        self.tokenizer = tokenizer          # resolved lazily: the merge table is data outside this repository
        self._clip_model = clip_model
        self.dtype = {"bfloat16": torch.bfloat16, "float16": torch.float16}.get(dtype, torch.float32)
        self.net.text.compute_dtype = self.dtype          # float16: the f16 flavour of the kernels (the reference's eval dtype)
        if cache_dir:
            self.load_pretrained(cache_dir)
        if freeze:
            self.net.eval()
            for p in self.net.parameters():
                p.requires_grad = False
        self.register_buffer("unused_tensor", torch.zeros(1), False)
        self.text_list_to_feature = {}
        self.max_batch_size = max_batch_size
        self.all_positions = all_positions

    def load_pretrained(self, path):
        """EVA-CLIP checkpoint (`EVA02_CLIP_E_psz14_plus_s9B.pt`): a flat state dict, `text.*` + `logit_scale` (+ `visual.*`,
        ignored: the wrapper deletes the visual tower, :29)"""
        sd = torch.load(path, map_location="cpu")
        for k in ("state_dict", "model", "module"):
            if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
                sd = sd[k]
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
        own = self.net.state_dict()
        sd = {k: v for k, v in sd.items() if k in own}
        missing = [k for k in own if k not in sd]
        if missing:
            raise RuntimeError(f"EVA02CLIP.load_pretrained: {len(missing)} text-tower keys missing from {path}, e.g. {missing[:3]}")
        self.net.load_state_dict(sd)

    @property
    def device(self):
        return self.unused_tensor.device

    def _tokens(self, text_list):
        if self.tokenizer is None:
            self.tokenizer = get_tokenizer(self._clip_model)
        return self.tokenizer(text_list, context_length=self.net.text.context_length).to(self.device)

    @torch.no_grad()
    def forward_tokens(self, text_token):
        """token ids [K, 77] (device) -> the forward_text dict"""
        K, ctx = text_token.shape
        feats, fulls = [], []
        for s in range(0, K, self.max_batch_size):
            f, full, L = self.net.text.forward_tokens(text_token[s:s + self.max_batch_size], all_positions=True if self.all_positions else False)
            feats.append(f)
            fulls.append(full)
        end_token_idx = text_token.argmax(dim=-1)
        attention_mask = (torch.arange(ctx, device=text_token.device)[None, :] <= end_token_idx[:, None]).to(end_token_idx.dtype)
        ret = _TextFeatures({"end_token_idx": end_token_idx, "attention_mask": attention_mask, "last_hidden_state_eot": torch.cat(feats, 0)})
        if self.all_positions:
            ret["last_hidden_state"] = torch.cat(fulls, 0)
        else:
            # the reference always returns the projected features of all 77 positions (clip_wrapper_eva02.py:117-122); nothing on the
            # name-prompt path reads them, so they are computed on first access instead of on every call (7x the work of the
            # end-of-text features on the truncated context)
            ret._lazy_full = lambda: self._all_position_features(text_token)
        return ret

    @torch.no_grad()
    def _all_position_features(self, text_token):
        fulls = [self.net.text.forward_tokens(text_token[s:s + self.max_batch_size], all_positions=True)[1]
                 for s in range(0, text_token.shape[0], self.max_batch_size)]
        return torch.cat(fulls, 0)

    @torch.no_grad()
    def forward_text(self, text_list, cache=False):
        key = tuple(text_list)
        if cache and key in self.text_list_to_feature:
            return self.text_list_to_feature[key]
        ret = self.forward_tokens(self._tokens(list(text_list)))
        if cache:
            self.text_list_to_feature[key] = ret
        return ret

    @torch.no_grad()
    def encode_text(self, text_list, cache=False):
        """(:54-84) the end-of-text features only"""
        return {"last_hidden_state_eot": self.forward_text(text_list, cache=cache)["last_hidden_state_eot"]}
