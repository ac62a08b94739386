"""HipLateFusion: the network of the reference's LateFusionModel (st_water_seg/models/lf_model.py:29-92,
feat_fusion='concat_conv') executed by libfloodunet.so -- one UNetEncoder per input (unet.py:134-159), the per-level
concat + Conv2d(n*fs, fs, 1) fusion (lf_model.py:40-45, 78-90) and one UNetDecoder (unet.py:162-188).

state_dict keys are the reference's: ``encoders.<input name>.inc.double_conv.0.weight`` ...,
``decoder.up1.conv.double_conv.0.weight`` ... ``decoder.outc.conv.bias``, ``concat_convs.<level>.weight|bias``.
The input is ONE tensor [B, sum(channels), H, W] holding the inputs side by side in the order the reference's forward
concatenates their features (lf_model.py:58-76: image, dem, slope, preflood, pre_post_difference, hand); the C side
gives every encoder its channel window.  Everything else (flat storage, fused loss, autograd glue, Adam, data-parallel
trainer) is HipUNet's."""
from __future__ import annotations

from typing import Dict, List

import torch.nn as nn

from .unet import HipUNet, _ConvParams, _Holder, _build_decoder, _build_encoder, channel_plan

# lf_model.py:58-76: the order in which forward() appends the encoders' features
FORWARD_ORDER = ["ms_image", "dem", "slope", "preflood", "pre_post_difference", "hand"]


class HipLateFusion(HipUNet):
    def __init__(self, in_channels: Dict[str, int], n_classes: int, base_channels: int = 64, precision: str = "fp32"):
        if not isinstance(in_channels, dict) or len(in_channels) < 1:
            raise ValueError("in_channels must be a non-empty dict {input name: channels}")
        unknown = [k for k in in_channels if k not in FORWARD_ORDER]
        if unknown:
            raise ValueError(f"LateFusionModel.forward knows the inputs {FORWARD_ORDER}; got {unknown}")
        if "ms_image" not in in_channels:
            raise KeyError("ms_image")                       # lf_model.py:56 indexes self.encoders['ms_image']
        if len(in_channels) > 6:
            raise ValueError("at most 6 encoders")
        self._in_channels = dict(in_channels)
        self.encoder_names: List[str] = [k for k in FORWARD_ORDER if k in in_channels]
        super().__init__(sum(in_channels.values()), n_classes, bilinear=True, base_channels=base_channels,
                         precision=precision)

    def _build_tree(self):
        base = self.base_channels
        # construction (= RNG) order of the reference: encoders in in_channels order, decoder, concat_convs
        # (lf_model.py:31-45); registration order: encoders in forward order, concat_convs, decoder -- the flat
        # parameter buffer then runs encoders | fusion | decoder, adjacent in backward order for the gradient buckets
        built = {}
        for name, ch in self._in_channels.items():
            h = _Holder()
            _build_encoder(h, ch, base, True)
            built[name] = h
        dec = _Holder()
        _build_decoder(dec, self.n_classes, base, True)
        feats, _ = channel_plan(1, base, True)
        cc = _Holder()
        n = len(self._in_channels)
        for l, fs in enumerate(feats):
            cc.add_module(str(l), _ConvParams(fs * n, fs, 1))
        enc = _Holder()
        for name in self.encoder_names:
            enc.add_module(name, built[name])
        self.encoders = enc
        self.concat_convs = cc
        self.decoder = dec

    def _enc_config(self):
        return len(self.encoder_names), [self._in_channels[k] for k in self.encoder_names]

    def _c_param_name(self, name: str) -> str:
        if name.startswith("encoders."):
            _, key, rest = name.split(".", 2)
            return f"encoders.{self.encoder_names.index(key)}.{rest}"
        return name

    def channel_windows(self) -> Dict[str, slice]:
        """input name -> channel slice of the concatenated input tensor."""
        out, off = {}, 0
        for k in self.encoder_names:
            out[k] = slice(off, off + self._in_channels[k])
            off += self._in_channels[k]
        return out
