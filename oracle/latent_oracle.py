"""Functional CPU restatement of the latent path: SDUNet_CAR and VAE_InvTrans (test oracle; see oracle/__init__.py).

PARITY UNPINNED.  The reference holds no test, golden vector or saved output for these models, and the reference itself is
study-only in this pipeline, so this file was written by reading the sources it cites and is checked
only for self-consistency (shapes, the documented decode-length bug, invariances).  Every stage is a stock torch CPU op
(F.conv1d, F.group_norm, F.multi_head_attention_forward, F.interpolate, F.batch_norm, F.adaptive_avg_pool1d): the same
ATen kernels the reference's nn.Modules dispatch to, so the restatement can only differ from the reference where the reading
of the module graph is wrong.

Works on a plain ``state_dict`` with the reference's key names.  Dropout is the identity (eval mode, eval_sd.ipynb).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def swish(x: torch.Tensor) -> torch.Tensor:
    # Models/ModelUtils.py:13-14
    return x * torch.sigmoid(x)


def positional_table(max_time: int, dim: int, dtype=torch.float32) -> torch.Tensor:
    # Models/StableDiffusionBlocks.py:92-98 (fp32 sin/cos interleaved over hidden_dim)
    pos = torch.arange(max_time, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(1.0e4) / dim))
    pe = torch.zeros((max_time, dim), dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def condition_mix(sd: SD, time: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor, max_time: int,
                  prefix: str = "embed_block.") -> torch.Tensor:
    """ConditionMixBlock.forward, Models/StableDiffusionBlocks.py:124-142.  Returns (B, 7 * embed_dim)."""
    p = prefix
    w0 = sd[p + "time_embed_layers.0.weight"]
    pe = positional_table(max_time, w0.shape[1], w0.dtype)
    h = F.leaky_relu(F.linear(pe[time, :], w0, sd[p + "time_embed_layers.0.bias"]), 0.1)
    time_embed = F.linear(h, sd[p + "time_embed_layers.2.weight"], sd[p + "time_embed_layers.2.bias"])
    cell = F.embedding(cat_attr[:, :-1], sd[p + "cell_index_embedder.weight"]).flatten(1)
    day = F.embedding(cat_attr[:, -1], sd[p + "day_embedder.weight"])
    cat_embed = torch.cat([cell, day], dim=1)   # :136 as is: `categorical_fc` (:109) is constructed but never called by forward
    num_embed = F.linear(num_attr.to(w0.dtype), sd[p + "numerical_fc.weight"], sd[p + "numerical_fc.bias"])
    o = torch.cat([time_embed, cat_embed, num_embed], dim=1)
    o = F.linear(F.leaky_relu(o, 0.1), sd[p + "mix_fc.1.weight"], sd[p + "mix_fc.1.bias"])
    o = F.linear(F.leaky_relu(o, 0.1), sd[p + "mix_fc.3.weight"], sd[p + "mix_fc.3.bias"])
    return swish(o)


def mha(sd: SD, p: str, q: torch.Tensor, kv: torch.Tensor, heads: int) -> torch.Tensor:
    """nn.MultiheadAttention(E, heads, batch_first=True)(q, kv, kv)[0] in eval mode: what the module's forward computes
    (torch/nn/modules/activation.py: transpose to sequence-first, F.multi_head_attention_forward, transpose back)."""
    e = q.shape[-1]
    out, _ = F.multi_head_attention_forward(
        q.transpose(0, 1), kv.transpose(0, 1), kv.transpose(0, 1), e, heads, sd[p + "in_proj_weight"], sd[p + "in_proj_bias"], None, None,
        False, 0.0, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"], training=False, need_weights=False)
    return out.transpose(0, 1)


def cross_attention_res(sd: SD, p: str, x: torch.Tensor, cond: torch.Tensor, heads: int = 8, groups: int = 32) -> torch.Tensor:
    """CrossAttentionRes.forward, Models/StableDiffusionBlocks.py:69-82 (GN eps 1e-6, :48,:61)."""
    identity = F.conv1d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"]) if (p + "shortcut.weight") in sd else x
    h = F.group_norm(x, groups, sd[p + "stage1.0.weight"], sd[p + "stage1.0.bias"], eps=1e-6)
    h = F.conv1d(swish(h), sd[p + "stage1.2.weight"], sd[p + "stage1.2.bias"], padding=1).transpose(1, 2)   # (B, L, out)
    c = F.linear(cond, sd[p + "cond_proj.0.weight"], sd[p + "cond_proj.0.bias"]).unflatten(1, (7, -1))   # (B, 7, out)
    h = mha(sd, p + "multihead_attn.", h, c, heads) + h
    h = h.transpose(1, 2)
    h = F.group_norm(h, groups, sd[p + "stage2.0.weight"], sd[p + "stage2.0.bias"], eps=1e-6)
    h = F.conv1d(swish(h), sd[p + "stage2.2.weight"], sd[p + "stage2.2.bias"], padding=1)
    return identity + h


def sdunet_car_forward(sd: SD, x: torch.Tensor, time: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor,
                       channel_schedule: Sequence[int], diffusion_steps: int, res_blocks: int) -> torch.Tensor:
    """SDUNet_CAR.forward, Models/StableDiffusion.py:208-222, with the stage layouts of :154-171 (encoder: res_blocks
    CrossAttentionRes then the stride-2 conv; decoder: res_blocks - 1 CrossAttentionRes, nearest x2, one more)."""
    stages = len(channel_schedule) - 1
    emb = condition_mix(sd, time, cat_attr, num_attr, diffusion_steps)
    x = F.conv1d(x, sd["stem.weight"], sd["stem.bias"], padding=1)
    downs: List[torch.Tensor] = []
    for i in range(stages):                                   # :181-187
        for j in range(res_blocks):
            x = cross_attention_res(sd, f"down_blocks.{i}.{j}.", x, emb)
        x = F.conv1d(x, sd[f"down_blocks.{i}.{res_blocks}.weight"], sd[f"down_blocks.{i}.{res_blocks}.bias"], stride=2, padding=1)
        downs.append(x)
    x = cross_attention_res(sd, "mid_attn_block.", downs[-1], emb)   # :220
    for i in range(stages):                                   # :198-205
        x = torch.cat([x, downs[-i - 1]], dim=1)
        for j in range(res_blocks - 1):
            x = cross_attention_res(sd, f"up_blocks.{i}.{j}.", x, emb)
        x = F.interpolate(x, scale_factor=2, mode="nearest")   # nn.Upsample(scale_factor=2, mode='nearest'), :166
        x = cross_attention_res(sd, f"up_blocks.{i}.{res_blocks}.", x, emb)
    x = F.group_norm(x, 32, sd["head.0.weight"], sd["head.0.bias"], eps=1e-5)    # :148-151 (default eps, no activation)
    return F.conv1d(x, sd["head.1.weight"], sd["head.1.bias"], padding=1)


# ---- VAE_InvTrans (Models/VAE.py:135-192) ---------------------------------------------------------------------------------

def conv_norm_act(sd: SD, p: str, x: torch.Tensor, stride: int = 1) -> torch.Tensor:
    """ConvNormAct(in, out, 3, stride, 1): Conv1d without bias, BatchNorm1d (eval: running statistics), Swish
    (Models/ModelUtils.py:17-33)."""
    h = F.conv1d(x, sd[p + "0.weight"], None, stride=stride, padding=1)
    h = F.batch_norm(h, sd[p + "1.running_mean"], sd[p + "1.running_var"], sd[p + "1.weight"], sd[p + "1.bias"], False, 0.1, 1e-5)
    return swish(h)


def inv_transformer_block(sd: SD, p: str, x: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """InvTransformerBlock.forward, Models/VAEBlocks.py:118-121: attention ACROSS CHANNELS (tokens = channels, embedding =
    sequence positions), then a conv residual."""
    x = mha(sd, p + "attn.", x, x, heads) + x
    h = conv_norm_act(sd, p + "layers.0.", x)
    h = F.conv1d(h, sd[p + "layers.1.weight"], sd[p + "layers.1.bias"], padding=1)
    return swish(x + h)


def vae_encode(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """VAE_InvTrans.encode, Models/VAE.py:174-182 with the encoder of :141-153.  (B, 2, L) -> (B, 4, L // 8)."""
    L = x.shape[-1]
    h = F.interpolate(x, size=256, mode="linear", align_corners=True)
    h = conv_norm_act(sd, "encoder.1.", h)
    for k, stride in ((2, 2), (4, 2), (6, 2)):
        h = conv_norm_act(sd, f"encoder.{k}.", h, stride)
        h = inv_transformer_block(sd, f"encoder.{k + 1}.", h)
    h = F.adaptive_avg_pool1d(h, L // 8)
    return F.conv1d(h, sd["mu_head.weight"], sd["mu_head.bias"])


def vae_decode(sd: SD, z: torch.Tensor) -> torch.Tensor:
    """VAE_InvTrans.decode, Models/VAE.py:185-192 with the decoder of :155-172.  The pooled length is taken from the LATENT
    (`_, _, L = z.shape`), so (B, 4, l) -> (B, 2, l), not (B, 2, 8 l): the reference's behaviour, restated as is."""
    l = z.shape[-1]
    h = F.interpolate(z, size=32, mode="linear", align_corners=True)
    h = conv_norm_act(sd, "decoder.1.", h)
    for k in (2, 5, 8):
        h = inv_transformer_block(sd, f"decoder.{k}.", h)
        h = F.interpolate(h, scale_factor=2, mode="linear", align_corners=True)
        h = conv_norm_act(sd, f"decoder.{k + 2}.", h)
    h = F.conv1d(h, sd["decoder.11.weight"], sd["decoder.11.bias"], padding=1)
    return F.adaptive_avg_pool1d(h, l)
