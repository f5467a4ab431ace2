"""AC-SASRec with the reference's model surface, running on the MI355X-native encoder.

Mirrors recbole/model/abstract_recommender.py:108-143 (SequentialRecommender) and
recbole/model/sequential_recommender/acsasrec.py:9-164 (ACSASRec): same constructor arguments
(`config` mapping + `dataset.num(field)`), attribute names, method names, return values and
state-dict keys.  `config[k]` may be any mapping; missing keys read as None like RecBole's Config
(recbole/config/configurator.py:405-409).
"""
from __future__ import annotations

import logging

import random
from enum import Enum

import torch
from torch import nn

from . import layers
from .layers import AttackRTransformerEncoder
from . import ce, fused_embed
from .linear import embedding_lookup, full_sort_scores
from .ops import StructuredMask, mask_penalty
from .state import StepState


class ModelType(Enum):
    """recbole/utils/enum_type.py (only the member this path needs)."""

    SEQUENTIAL = 2


def _cfg(config, key, default=None):
    try:
        val = config[key]
    except KeyError:
        val = None
    return default if val is None else val


def _penalty(attack_mask, mode="local"):
    """torch.norm(1 - attack_mask, p=2) (acsasrec.py:135, acbert4rec.py:231).  mode 'global' (config key
    `dp_mask_penalty`): the norm over every data-parallel rank's batch (parallel.global_mask_penalty)."""
    if mode == "global":
        from .parallel import global_mask_penalty
        return global_mask_penalty(attack_mask)
    pen = getattr(attack_mask, "_acattn_pen", None)  # the attention node's row sums of (1 - M)^2 (ops.PENALTY_ROWS)
    if pen is not None:
        from . import ops
        if ops.PENALTY_ROWS:
            # gradient: a [B, nh, ceil(L/16)] cotangent into the attention backward, no dense d M.  (A zero norm -- M == 1
            # everywhere, i.e. L = 1 -- gets a zero gradient like torch.norm's, not the square root's infinity.)
            return torch.sqrt(pen.sum().clamp_min(1e-30))
    if attack_mask.is_cuda and attack_mask.dtype == torch.float32:
        return mask_penalty(attack_mask)
    return torch.norm(1 - attack_mask, p=2)


def _front_end(model, item_seq, keep_emb=None, bidirectional=False):
    """(dropout(LayerNorm(item_embedding(item_seq) + position_embedding)), structured mask): acsasrec.py:87-98 ==
    acbert4rec.py:163-173.  One fused launch each way on the HIP path (fused_embed), which also writes the mask's
    key-validity bytes (item_seq != 0); hidden sizes it does not cover take the same chain as separate device ops."""
    pos = model.position_embedding if model.use_position_embedding else None
    if pos is not None and item_seq.size(1) > pos.num_embeddings:
        raise IndexError("index out of range in self")  # what nn.Embedding raises on the host
    if item_seq.is_cuda and fused_embed.supported(model.hidden_size):
        emb, nonzero = fused_embed.embed_layer_norm(item_seq, model.item_embedding, pos, model.LayerNorm, model.dropout.p,
                                                    model.training, keep_emb, return_nonzero=True)
        return emb, StructuredMask(key_valid=nonzero, causal=not bidirectional)
    input_emb = embedding_lookup(item_seq, model.item_embedding)
    if pos is not None:
        position_ids = torch.arange(item_seq.size(1), dtype=torch.long, device=item_seq.device)
        input_emb = input_emb + pos(position_ids).unsqueeze(0)
    input_emb = model.LayerNorm(input_emb)
    mask = model.get_structured_mask(item_seq, bidirectional)
    if keep_emb is not None:
        return input_emb * (keep_emb.to(input_emb.dtype) / (1.0 - model.dropout.p)), mask
    return model.dropout(input_emb), mask


class SequentialRecommender(nn.Module):
    """recbole/model/abstract_recommender.py:108-143."""

    type = ModelType.SEQUENTIAL

    def __init__(self, config, dataset):
        super().__init__()
        self.USER_ID = _cfg(config, 'USER_ID_FIELD', 'user_id')
        self.ITEM_ID = _cfg(config, 'ITEM_ID_FIELD', 'item_id')
        self.ITEM_SEQ = self.ITEM_ID + _cfg(config, 'LIST_SUFFIX', '_list')
        self.ITEM_SEQ_LEN = _cfg(config, 'ITEM_LIST_LENGTH_FIELD', 'item_length')
        self.POS_ITEM_ID = self.ITEM_ID
        self.NEG_ITEM_ID = _cfg(config, 'NEG_PREFIX', 'neg_') + self.ITEM_ID
        self.max_seq_length = _cfg(config, 'MAX_ITEM_LIST_LENGTH', 50)
        self.n_items = dataset.num(self.ITEM_ID)
        self.device = _cfg(config, 'device', 'cuda')

    def gather_indexes(self, output, gather_index):
        """Gathers the vectors at the specific positions over a minibatch (abstract_recommender.py:130-134)."""
        gather_index = gather_index.view(-1, 1, 1).expand(-1, -1, output.shape[-1])
        return output.gather(dim=1, index=gather_index).squeeze(1)

    def get_attention_mask(self, item_seq, bidirectional=False):
        """Dense additive mask exactly as the reference builds it (abstract_recommender.py:136-143)."""
        attention_mask = (item_seq != 0)
        extended_attention_mask = attention_mask.unsqueeze(1).unsqueeze(2)
        if not bidirectional:
            extended_attention_mask = torch.tril(extended_attention_mask.expand((-1, -1, item_seq.size(-1), -1)))
        return torch.where(extended_attention_mask, 0., -10000.)

    def get_structured_mask(self, item_seq, bidirectional=False) -> StructuredMask:
        """Same mask in factored form: 1 byte per key instead of L*L floats per sequence."""
        return StructuredMask(key_valid=(item_seq != 0).to(torch.uint8), causal=not bidirectional)


class ACSASRec(SequentialRecommender):
    """recbole/model/sequential_recommender/acsasrec.py:9-164."""

    bidirectional = False

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.n_layers = config['n_layers']
        self.n_heads = config['n_heads']
        self.hidden_size = config['hidden_size']
        self.inner_size = config['inner_size']
        self.hidden_dropout_prob = config['hidden_dropout_prob']
        self.attn_dropout_prob = config['attn_dropout_prob']
        self.hidden_act = config['hidden_act']
        self.layer_norm_eps = config['layer_norm_eps']
        self.initializer_range = config['initializer_range']
        self.loss_type = config['loss_type']
        self.combine_option = config['combine_option']
        self.rich_calibrated_combine = _cfg(config, 'rich_calibrated_combine')
        self.two_level = _cfg(config, 'two_level')
        self.use_position_embedding = _cfg(config, 'use_position_embedding')
        self.use_order = _cfg(config, 'use_order')
        self.use_distance = _cfg(config, 'use_distance')
        self.trainable_mask_loss_weight = _cfg(config, 'trainable_mask_loss_weight')
        # not a reference key: how the mask penalty is taken under batch data-parallelism ('local': per-rank norm, the
        # DDP convention; 'global': the norm one process on the concatenated batch would see -- parallel.py)
        self.dp_mask_penalty = _cfg(config, 'dp_mask_penalty') or 'local'
        assert self.dp_mask_penalty in ('local', 'global'), self.dp_mask_penalty
        # The reference never forwards seq_length (acsasrec.py:40-54), which pins the gate to L = 50
        # (layers.py:863,878).  `gate_seq_length` is an opt-in extension for other lengths.
        seq_length = _cfg(config, 'gate_seq_length', 50)

        self.item_embedding = nn.Embedding(self.n_items, self.hidden_size, padding_idx=0)
        if self.use_position_embedding:
            self.position_embedding = nn.Embedding(self.max_seq_length, self.hidden_size)
        self.trm_encoder = AttackRTransformerEncoder(
            n_layers=self.n_layers, n_heads=self.n_heads, hidden_size=self.hidden_size, inner_size=self.inner_size,
            hidden_dropout_prob=self.hidden_dropout_prob, attn_dropout_prob=self.attn_dropout_prob,
            hidden_act=self.hidden_act, layer_norm_eps=self.layer_norm_eps, combine_option=self.combine_option,
            use_order=self.use_order, use_distance=self.use_distance, two_level=self.two_level,
            rich_calibrated_combine=self.rich_calibrated_combine, seq_length=seq_length)
        self.LayerNorm = nn.LayerNorm(self.hidden_size, eps=self.layer_norm_eps)
        self.dropout = nn.Dropout(self.hidden_dropout_prob)
        if self.trainable_mask_loss_weight:
            self.mask_loss_weight = nn.Parameter(torch.FloatTensor([0.3]), requires_grad=True)
        else:
            self.mask_loss_weight = config['mask_loss_weight']
        if self.loss_type == 'BPR':
            self.loss_fct = _BPRLoss()
        elif self.loss_type == 'CE':
            self.loss_fct = nn.CrossEntropyLoss()
        else:
            raise NotImplementedError("Make sure 'loss_type' in ['BPR', 'CE']!")
        self.apply(self._init_weights)
        # pass identity of the two-pass trainer, replay seed counter, schedule switch: per model (state.py)
        self.step_state = StepState().attach(self)

    def _init_weights(self, module):
        """acsasrec.py:74-84: N(0, initializer_range) weights, zero biases, unit LayerNorm."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def forward(self, item_seq, item_seq_len, is_train=False, _rnds=None, _keep_emb=None, _last_row=None):
        """`_last_row` ([B] int64, optional): item_seq_len - 1 already on the device (the trainer's graph mode forms it in
        the launch that copies the batch in: acattn_step_inputs)."""
        input_emb, mask = _front_end(self, item_seq, _keep_emb, self.bidirectional)
        # only position item_seq_len - 1 of the last layer is read (acsasrec.py:100-103): the encoder is told, so the
        # last layer's position-wise tail runs on B rows instead of B * L (gather and tail commute)
        if not self.step_state.prune_dead_work:  # the reference's full schedule (acsasrec.py:99-103)
            trm_output = self.trm_encoder(input_emb, mask, output_all_encoded_layers=True, _rnds=_rnds)
            attacked_output, calibrated_output = trm_output[0][-1]
            return (self.gather_indexes(attacked_output, item_seq_len - 1),
                    self.gather_indexes(calibrated_output, item_seq_len - 1), trm_output[1])
        last = (item_seq_len - 1).view(-1, 1) if _last_row is None else _last_row.view(-1, 1)
        trm_output = self.trm_encoder(input_emb, mask, output_all_encoded_layers=False, _rnds=_rnds, _last_rows=last)
        all_attack_masks = trm_output[1]
        attacked_output, calibrated_output = trm_output[0][-1]
        return attacked_output.squeeze(1), calibrated_output.squeeze(1), all_attack_masks

    def _cal_loss(self, output, interaction, attack_loss=False):
        pos_items = interaction[self.POS_ITEM_ID]
        if self.loss_type == 'BPR':
            neg_items = interaction[self.NEG_ITEM_ID]
            pos_score = torch.sum(output * self.item_embedding(pos_items), dim=-1)
            neg_score = torch.sum(output * self.item_embedding(neg_items), dim=-1)
            return self.loss_fct(pos_score, neg_score)
        if output.is_cuda and ce.supported(self.hidden_size) and torch.is_grad_enabled():
            # fused: the [B, n_items] logits (205 MB at 512 x 100k) are never written
            # the attacked loss is differentiated for the attack transforms only (trainer.py:678-684): no table gradient
            return ce.full_sort_cross_entropy(output, self.item_embedding.weight, pos_items, table_grad=not attack_loss,
                                              state=self.step_state)
        logits = full_sort_scores(output, self.item_embedding.weight, self.step_state)
        return self.loss_fct(logits, pos_items)

    def calculate_loss(self, interaction, _rnds=None, _keep_emb=None):
        item_seq = interaction[self.ITEM_SEQ]
        item_seq_len = interaction[self.ITEM_SEQ_LEN]
        last_row = interaction.get("_acattn_last_row") if isinstance(interaction, dict) else None
        attacked_output, calibrated_output, all_attack_masks = self.forward(item_seq, item_seq_len, is_train=True,
                                                                            _rnds=_rnds, _keep_emb=_keep_emb, _last_row=last_row)
        final_attacked_loss = None
        if attacked_output is not None:
            if (self.loss_type == 'CE' and not self.trainable_mask_loss_weight and attacked_output.is_cuda
                    and ce.supported(self.hidden_size) and torch.is_grad_enabled() and self.dp_mask_penalty == 'local'):
                # the whole expression below as one node (CE sweep with its direction, one finishing launch)
                final_attacked_loss = ce.attacked_loss(attacked_output, self.item_embedding.weight,
                                                       interaction[self.POS_ITEM_ID], all_attack_masks,
                                                       self.mask_loss_weight, self.step_state)
        if attacked_output is not None and final_attacked_loss is None:
            attacked_loss = -self._cal_loss(attacked_output, interaction, attack_loss=True)
            mask_penalty = [_penalty(m, self.dp_mask_penalty) for m in all_attack_masks if m is not None]
            assert len(mask_penalty) > 0
            mask_penalty = torch.mean(torch.stack(mask_penalty, dim=0))
            if self.trainable_mask_loss_weight:
                final_attacked_loss = attacked_loss + mask_penalty * self.mask_loss_weight[0]
            else:
                final_attacked_loss = attacked_loss + mask_penalty * self.mask_loss_weight
        calibrated_loss = self._cal_loss(calibrated_output, interaction)
        return final_attacked_loss, calibrated_loss

    def predict(self, interaction):
        item_seq = interaction[self.ITEM_SEQ]
        item_seq_len = interaction[self.ITEM_SEQ_LEN]
        test_item = interaction[self.ITEM_ID]
        attacked_output, calibrated_output, _ = self.forward(item_seq, item_seq_len)
        test_item_emb = self.item_embedding(test_item)
        attacked_scores = torch.mul(attacked_output, test_item_emb).sum(dim=1)
        scores = torch.mul(calibrated_output, test_item_emb).sum(dim=1)
        return attacked_scores, scores

    def full_sort_predict(self, interaction, _rnds=None):
        item_seq = interaction[self.ITEM_SEQ]
        item_seq_len = interaction[self.ITEM_SEQ_LEN]
        _, calibrated_output, _ = self.forward(item_seq, item_seq_len, _rnds=_rnds)
        scores = torch.matmul(calibrated_output, self.item_embedding.weight.transpose(0, 1))
        return None, scores


class AcBERT4Rec(SequentialRecommender):
    """recbole/model/sequential_recommender/acbert4rec.py:11-267: cloze (masked-item) training over the same
    calibrated encoder with a bidirectional mask.

    Same constructor, methods, return values and state-dict keys.  Two additions, both opt-in through config keys
    the reference does not know: `cloze_on_device` builds the masked batch with tensor ops on the device (the
    reference walks Python lists on the host, acbert4rec.py:105-150) and `gate_seq_length` as in ACSASRec.
    Reference behaviour kept on purpose: `masked_index > 0` marks real slots, so a masked FIRST position does not
    contribute to the loss (acbert4rec.py:226); evaluation appends a mask token (L+1 columns), which the gate
    (width pinned to L) and the position table (L rows) cannot take (acbert4rec.py:47,152-160)."""

    bidirectional = True

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.n_layers = config['n_layers']
        self.n_heads = config['n_heads']
        self.hidden_size = config['hidden_size']
        self.inner_size = config['inner_size']
        self.hidden_dropout_prob = config['hidden_dropout_prob']
        self.attn_dropout_prob = config['attn_dropout_prob']
        self.hidden_act = config['hidden_act']
        self.layer_norm_eps = config['layer_norm_eps']
        self.mask_ratio = config['mask_ratio']
        self.loss_type = config['loss_type']
        self.initializer_range = config['initializer_range']
        self.combine_option = config['combine_option']
        self.rich_calibrated_combine = _cfg(config, 'rich_calibrated_combine')
        self.two_level = _cfg(config, 'two_level')
        self.use_position_embedding = _cfg(config, 'use_position_embedding')
        self.use_order = _cfg(config, 'use_order')
        self.use_distance = _cfg(config, 'use_distance')
        self.trainable_mask_loss_weight = _cfg(config, 'trainable_mask_loss_weight')
        # not a reference key: how the mask penalty is taken under batch data-parallelism ('local': per-rank norm, the
        # DDP convention; 'global': the norm one process on the concatenated batch would see -- parallel.py)
        self.dp_mask_penalty = _cfg(config, 'dp_mask_penalty') or 'local'
        assert self.dp_mask_penalty in ('local', 'global'), self.dp_mask_penalty
        self.cloze_on_device = bool(_cfg(config, 'cloze_on_device', False))
        seq_length = _cfg(config, 'gate_seq_length', 50)

        self.mask_token = self.n_items
        self.mask_item_length = int(self.mask_ratio * self.max_seq_length)
        self.item_embedding = nn.Embedding(self.n_items + 1, self.hidden_size, padding_idx=0)  # + the mask token
        # a fifth of all positions look up the mask token (mask_ratio): the fused embedding backward sums that row's
        # gradient inside each workgroup instead of adding 20k rows to it one atomic at a time (acattn_embed_problem.hot_id_plus1)
        self.item_embedding._acattn_hot_id = self.mask_token
        if self.use_position_embedding:
            self.position_embedding = nn.Embedding(self.max_seq_length, self.hidden_size)
        self.trm_encoder = AttackRTransformerEncoder(
            n_layers=self.n_layers, n_heads=self.n_heads, hidden_size=self.hidden_size, inner_size=self.inner_size,
            hidden_dropout_prob=self.hidden_dropout_prob, attn_dropout_prob=self.attn_dropout_prob,
            hidden_act=self.hidden_act, layer_norm_eps=self.layer_norm_eps, combine_option=self.combine_option,
            use_order=self.use_order, use_distance=self.use_distance, two_level=self.two_level,
            rich_calibrated_combine=self.rich_calibrated_combine, seq_length=seq_length)
        self.LayerNorm = nn.LayerNorm(self.hidden_size, eps=self.layer_norm_eps)
        self.dropout = nn.Dropout(self.hidden_dropout_prob)
        if self.trainable_mask_loss_weight:
            self.mask_loss_weight = nn.Parameter(torch.FloatTensor([0.3]), requires_grad=True)
        else:
            self.mask_loss_weight = config['mask_loss_weight']
        if self.loss_type not in ('BPR', 'CE'):
            raise AssertionError("Make sure 'loss_type' in ['BPR', 'CE']!")
        self.apply(self._init_weights)
        # pass identity of the two-pass trainer, replay seed counter, schedule switch: per model (state.py)
        self.step_state = StepState().attach(self)

    _init_weights = ACSASRec._init_weights

    # ---- cloze reconstruction ---------------------------------------------------------------------------------
    def _neg_sample(self, item_set):
        item = random.randint(1, self.n_items - 1)
        while item in item_set:
            item = random.randint(1, self.n_items - 1)
        return item

    def _padding_sequence(self, sequence, max_length):
        return ([0] * (max_length - len(sequence)) + sequence)[-max_length:]

    def reconstruct_train_data(self, item_seq):
        """Masked sequence, positives, negatives and masked positions (acbert4rec.py:105-150).  The host flavour
        draws from Python's `random` in the reference's order, so a seeded `random` gives the reference's batch."""
        if self.cloze_on_device:
            return self._reconstruct_train_data_device(item_seq)
        rows = item_seq.cpu().tolist()
        masked_rows, pos_rows, neg_rows, idx_rows = [], [], [], []
        for row in rows:
            masked, pos, neg, idx = list(row), [], [], []
            for j, item in enumerate(row):
                if item == 0:  # right-padded: the sequence has ended
                    break
                if random.random() < self.mask_ratio:
                    pos.append(item)
                    neg.append(self._neg_sample(row))
                    masked[j] = self.mask_token
                    idx.append(j)
            masked_rows.append(masked)
            for dst, src in ((pos_rows, pos), (neg_rows, neg), (idx_rows, idx)):
                dst.append(self._padding_sequence(src, self.mask_item_length))
        to = lambda v: torch.tensor(v, dtype=torch.long, device=item_seq.device).view(len(rows), -1)
        return to(masked_rows), to(pos_rows), to(neg_rows), to(idx_rows)

    def _reconstruct_train_data_device(self, item_seq, generator=None):
        """The same batch distribution built with tensor ops on the device: no host round trip, no
        synchronisation (usable inside a captured graph).  A masked item keeps its order; the short lists are
        right-aligned and keep the LAST mask_item_length entries, like `_padding_sequence`."""
        B, L = item_seq.shape
        ml, dev = self.mask_item_length, item_seq.device
        real = (item_seq != 0).long().cumprod(dim=1).bool()  # the reference stops at the first padding
        m = real & (torch.rand(B, L, device=dev, generator=generator) < self.mask_ratio)
        masked_seq = torch.where(m, torch.full_like(item_seq, self.mask_token), item_seq)
        count = m.sum(dim=1, keepdim=True)
        slot = ml - count + (m.long().cumsum(dim=1) - 1)
        slot = torch.where(m & (slot >= 0), slot, torch.full_like(slot, ml))  # column ml = dropped
        # negatives: uniform over the catalogue, redrawn a few times where they hit an item of the sequence
        neg = torch.randint(1, self.n_items, (B, L), device=dev, generator=generator)
        for _ in range(8):
            clash = (neg.unsqueeze(2) == item_seq.unsqueeze(1)).any(dim=2)
            neg = torch.where(clash, torch.randint(1, self.n_items, (B, L), device=dev, generator=generator), neg)
        position = torch.arange(L, device=dev).expand(B, L)
        out = []
        for src in (item_seq, neg, position):
            buf = torch.zeros(B, ml + 1, dtype=torch.long, device=dev)
            out.append(buf.scatter_(1, slot, src)[:, :ml])
        pos_items, neg_items, masked_index = out
        return masked_seq, pos_items, neg_items, masked_index

    def reconstruct_test_data(self, item_seq, item_seq_len):
        """One more column, the mask token right after the last item (acbert4rec.py:152-160)."""
        pad = torch.zeros(item_seq.size(0), 1, dtype=torch.long, device=item_seq.device)
        item_seq = torch.cat((item_seq, pad), dim=-1)
        return item_seq.scatter(1, item_seq_len.view(-1, 1), self.mask_token)

    # ---- model ------------------------------------------------------------------------------------------------------
    def forward(self, item_seq, _rnds=None, _keep_emb=None, _rows=None):
        """(attacked [B,L,H], calibrated [B,L,H], attack masks); with `_rows` ([B,R] positions) only those positions
        of the two outputs, [B,R,H] (see AttackRTransformerLayer.forward)."""
        input_emb, mask = _front_end(self, item_seq, _keep_emb, bidirectional=True)
        if not self.step_state.prune_dead_work:  # the reference's full schedule, rows picked afterwards (acbert4rec.py:219-225)
            trm_output = self.trm_encoder(input_emb, mask, output_all_encoded_layers=True, _rnds=_rnds)
            attacked_output, calibrated_output = trm_output[0][-1]
            if _rows is not None:
                index = _rows.unsqueeze(-1).expand(-1, -1, attacked_output.size(-1))
                attacked_output, calibrated_output = attacked_output.gather(1, index), calibrated_output.gather(1, index)
            return attacked_output, calibrated_output, trm_output[1]
        trm_output = self.trm_encoder(input_emb, mask, output_all_encoded_layers=False, _rnds=_rnds, _last_rows=_rows)
        attacked_output, calibrated_output = trm_output[0][-1]
        return attacked_output, calibrated_output, trm_output[1]

    def multi_hot_embed(self, masked_index, max_length):
        """acbert4rec.py:180-199 (kept for callers; calculate_loss gathers the rows directly)."""
        masked_index = masked_index.view(-1)
        multi_hot = torch.zeros(masked_index.size(0), max_length, device=masked_index.device)
        multi_hot[torch.arange(masked_index.size(0)), masked_index] = 1
        return multi_hot

    def _cal_loss(self, seq_output, pos_items, targets, attack_loss=False):
        """CE over the catalogue without the mask-token row, averaged over the real masked slots
        (acbert4rec.py:201-209)."""
        table = self.item_embedding.weight[:self.n_items]
        rows = seq_output.reshape(-1, seq_output.size(-1))
        # hidden 256: the fused kernels run one item tile per wave there (acattn_ce.hip) and are MEASURED slower than
        # hipBLASLt + materialised logits for this model's ~20k masked slots (98.7 against 53.3 ms per step at 20k items,
        # round 3) -- they are taken when the [rows, N] logits would not be reasonable to hold (`ce_materialise_limit`
        # bytes per logits tensor, 16 GiB by default: 20k slots x 100k items is 8.2 GB and still materialises)
        limit = getattr(self, "ce_materialise_limit", 16 << 30)
        fused = rows.is_cuda and ce.supported(self.hidden_size) and torch.is_grad_enabled() and (
            self.hidden_size <= 128 or rows.shape[0] * table.shape[0] * 4 > limit)
        if fused:
            per_slot = ce.full_sort_cross_entropy_rows(rows, table, pos_items.reshape(-1), table_grad=not attack_loss,
                                                       state=self.step_state)
        else:
            # said once per process so that nobody has to find it in a profile
            if rows.is_cuda and not getattr(AcBERT4Rec, "_noted_materialised_ce", False):
                AcBERT4Rec._noted_materialised_ce = True
                logging.getLogger("ac_tsr_amd").info(
                    "AcBERT4Rec: hidden_size %d: the masked-slot CE uses materialised [%d, %d] logits (faster than the "
                    "fused kernels at this width; ce_materialise_limit switches)", self.hidden_size, rows.shape[0], table.shape[0])
            per_slot = ce.dense_cross_entropy_rows(full_sort_scores(rows, table, self.step_state), pos_items.reshape(-1))
        return torch.sum(per_slot * targets) / torch.sum(targets)

    def calculate_loss(self, interaction, _cloze=None, _rnds=None, _keep_emb=None):
        item_seq = interaction[self.ITEM_SEQ]
        masked_item_seq, pos_items, neg_items, masked_index = _cloze or self.reconstruct_train_data(item_seq)
        # the reference multiplies the outputs by a one-hot matrix (acbert4rec.py:219-225): a row gather, exactly;
        # the last layer's position-wise tail therefore only runs on the masked positions
        attacked_seq_output, calibrated_seq_output, all_attack_masks = self.forward(
            masked_item_seq, _rnds=_rnds, _keep_emb=_keep_emb, _rows=masked_index)
        targets = (masked_index > 0).float().view(-1)
        if self.loss_type == 'BPR':
            raise NotImplementedError("the reference computes only the CE loss here (acbert4rec.py:201-209)")
        attacked_loss = -self._cal_loss(attacked_seq_output, pos_items, targets, attack_loss=True)
        mask_penalty = torch.mean(torch.stack([_penalty(m, self.dp_mask_penalty) for m in all_attack_masks], dim=0))
        if self.trainable_mask_loss_weight:
            final_attacked_loss = attacked_loss + mask_penalty * self.mask_loss_weight[0]
        else:
            final_attacked_loss = attacked_loss + mask_penalty * self.mask_loss_weight
        calibrated_loss = self._cal_loss(calibrated_seq_output, pos_items, targets)
        return final_attacked_loss, calibrated_loss

    def predict(self, interaction):
        item_seq = self.reconstruct_test_data(interaction[self.ITEM_SEQ], interaction[self.ITEM_SEQ_LEN])
        item_seq_len = interaction[self.ITEM_SEQ_LEN]
        attacked_output, calibrated_output, _ = self.forward(item_seq, _rows=item_seq_len.view(-1, 1))
        test_item_emb = self.item_embedding(interaction[self.ITEM_ID])
        attacked_scores = torch.mul(attacked_output.squeeze(1), test_item_emb).sum(dim=1)
        scores = torch.mul(calibrated_output.squeeze(1), test_item_emb).sum(dim=1)
        return attacked_scores, scores

    def full_sort_predict(self, interaction, _rnds=None):
        item_seq_len = interaction[self.ITEM_SEQ_LEN]
        item_seq = self.reconstruct_test_data(interaction[self.ITEM_SEQ], item_seq_len)
        attacked_output, calibrated_output, _ = self.forward(item_seq, _rnds=_rnds, _rows=item_seq_len.view(-1, 1))
        test_items_emb = self.item_embedding.weight[:self.n_items]  # without the mask token
        attacked_scores = torch.matmul(attacked_output.squeeze(1), test_items_emb.transpose(0, 1))
        scores = torch.matmul(calibrated_output.squeeze(1), test_items_emb.transpose(0, 1))
        return attacked_scores, scores


class _BPRLoss(nn.Module):
    """recbole/model/loss.py:21-47."""

    def __init__(self, gamma=1e-10):
        super().__init__()
        self.gamma = gamma

    def forward(self, pos_score, neg_score):
        return -torch.log(self.gamma + torch.sigmoid(pos_score - neg_score)).mean()


class DictConfig(dict):
    """Plain-dict stand-in for RecBole's Config: missing keys read as None."""

    def __getitem__(self, k):
        return self.get(k, None)


class ItemCount:
    """Minimal `dataset` for the model constructors: only `.num(field)` is consulted."""

    def __init__(self, n_items):
        self.n_items = n_items

    def num(self, field):
        return self.n_items
