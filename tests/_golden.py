"""Loader for the committed golden fixtures (tests/golden/*.npz, written by oracle/gen_golden.py)."""
import os

import numpy as np
import torch

from oracle import ac_tsr_ref as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ENCODER_CASES = [
    "enc_gate_init", "enc_gate_stress", "enc_gate_h4", "enc_fixed_dist", "enc_fixed_order_bidir", "enc_gate_bidir",
    "enc_plain", "enc_onelevel", "enc_onelevel_trainable", "enc_anneal", "enc_leftpad", "enc_L200_h4", "enc_L200_d64_bidir", "enc_L37_ragged",
]
MODEL_CASES = ["model_eval", "model_eval_stress", "model_train", "model_beauty"]
BERT_CASES = ["bert_gate", "bert_fixed_scores"]


class Case:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.raw = {k: z[k] for k in z.files}

    def t(self, key):
        return torch.from_numpy(self.raw[key])

    def has(self, key):
        return key in self.raw

    def params(self):
        return {k[2:]: torch.from_numpy(v) for k, v in self.raw.items() if k.startswith("p.")}

    def grads(self):
        return {k[5:]: torch.from_numpy(v) for k, v in self.raw.items() if k.startswith("grad.")}

    def encoder_cfg(self):
        n_layers, h, H, inner, L, uo, ud, tl = (int(v) for v in self.raw["meta.cfg"])
        return O.EncoderCfg(n_layers=n_layers, n_heads=h, hidden_size=H, inner_size=inner, combine_option=str(
            self.raw["meta.combine"]), use_order=bool(uo), use_distance=bool(ud), two_level=bool(tl),
            rich_calibrated_combine=str(self.raw["meta.rich"]), seq_length=L)

    def model_cfg(self):
        n_layers, h, H, inner, L, n_items = (int(v) for v in self.raw["meta.cfg"])
        enc = O.EncoderCfg(n_layers=n_layers, n_heads=h, hidden_size=H, inner_size=inner,
                           combine_option=str(self.raw["meta.combine"]), rich_calibrated_combine="none", seq_length=50)
        return O.ModelCfg(enc=enc, n_items=n_items, max_seq_length=L,
                          mask_loss_weight=float(self.raw["meta.mask_loss_weight"]))

    def bert_cfg(self):
        """ModelCfg of an AcBERT4Rec fixture (bert_*.npz)."""
        cfg = self.model_cfg()
        cfg.bidirectional = True
        cfg.use_position_embedding = bool(int(self.raw["meta.use_pos"]))
        return cfg

    def layer_randomness(self, n_layers, train=False):
        out = []
        for i in range(n_layers):
            r = O.LayerRandomness(noise=self.t(f"in.noise.{i}"))
            if train:
                for f in ("keep_after", "keep_before", "keep_mask", "keep_out_att", "keep_out_cal", "keep_ffn_att",
                          "keep_ffn_cal"):
                    setattr(r, f, self.t(f"in.{f}.{i}").float())
            out.append(r)
        return out

    def batch(self):
        return {"item_id_list": self.t("in.item_id_list"), "item_length": self.t("in.item_length"),
                "item_id": self.t("in.item_id")}
