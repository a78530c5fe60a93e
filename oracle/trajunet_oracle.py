"""Functional CPU restatement of TrajUNet (test oracle; see oracle/__init__.py).

Works on a plain ``state_dict`` (name -> fp32 tensor) with the reference's key names, so
no nn.Module of the product or of the reference is involved.  Dropout is the identity
(inference; reference notebooks call ``.eval()``, eval_trajunet.ipynb:106).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def swish(x: torch.Tensor) -> torch.Tensor:
    # Models/ModelUtils.py:13-14
    return x * torch.sigmoid(x)


def positional_table(max_time: int, embed_dim: int = 128) -> torch.Tensor:
    # Models/TrajUNetBlocks.py:12-18 (fp32 sin/cos interleaved)
    pos = torch.arange(max_time, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, embed_dim, 2, dtype=torch.float32) * -(math.log(1.0e4) / embed_dim))
    pe = torch.zeros((max_time, embed_dim), dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def embed_forward(sd: SD, time: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor,
                  diffusion_steps: int, prefix: str = "embed_block.") -> torch.Tensor:
    """WideAndDeepEmbedBlock.forward, Models/TrajUNetBlocks.py:45-62.  Returns (B, 256, 1)."""
    p = prefix
    pe = positional_table(diffusion_steps)
    h = F.linear(pe[time, :], sd[p + "time_embed_layers.0.weight"], sd[p + "time_embed_layers.0.bias"])
    h = F.leaky_relu(h, 0.1)
    time_embed = F.linear(h, sd[p + "time_embed_layers.2.weight"], sd[p + "time_embed_layers.2.bias"])
    cell = F.embedding(cat_attr[:, :-1], sd[p + "cell_index_embedder.weight"]).flatten(1)
    day = F.embedding(cat_attr[:, -1], sd[p + "day_embedder.weight"])
    cat_embed = F.linear(torch.cat([cell, day], dim=1), sd[p + "categorical_fc.weight"], sd[p + "categorical_fc.bias"])
    num_embed = F.linear(num_attr, sd[p + "wide_fc.weight"], sd[p + "wide_fc.bias"])
    d = F.leaky_relu(torch.cat([cat_embed, num_embed], dim=1), 0.1)
    d = F.linear(d, sd[p + "deep_fc.1.weight"], sd[p + "deep_fc.1.bias"])
    d = F.leaky_relu(d, 0.1)
    attr_embed = F.linear(d, sd[p + "deep_fc.3.weight"], sd[p + "deep_fc.3.bias"])
    return swish(torch.cat([time_embed, attr_embed], dim=1).unsqueeze(2))


def resnet_block(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor, groups: int = 32) -> torch.Tensor:
    """ResnetBlock.forward, Models/TrajUNetBlocks.py:92-99 (GN eps 1e-6, :75,:83)."""
    e = F.conv1d(emb, sd[p + "embed_proj.weight"], sd[p + "embed_proj.bias"])
    h = F.group_norm(x, groups, sd[p + "stage1.0.weight"], sd[p + "stage1.0.bias"], eps=1e-6)
    h = F.conv1d(swish(h), sd[p + "stage1.2.weight"], sd[p + "stage1.2.bias"], padding=1)
    h = h + e
    h = F.group_norm(h, groups, sd[p + "stage2.0.weight"], sd[p + "stage2.0.bias"], eps=1e-6)
    h = F.conv1d(swish(h), sd[p + "stage2.3.weight"], sd[p + "stage2.3.bias"], padding=1)
    if (p + "shortcut.weight") in sd:
        x = F.conv1d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"])
    return x + h


def attn_block(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor, groups: int = 32) -> torch.Tensor:
    """AttnBlock.forward, Models/TrajUNetBlocks.py:117-125 (single head, scale C**-0.5)."""
    x = resnet_block(sd, p + "res1.", x, emb, groups)
    C = x.shape[1]
    h = F.group_norm(x, groups, sd[p + "qkv_getter.0.weight"], sd[p + "qkv_getter.0.bias"], eps=1e-6)
    qkv = F.conv1d(h, sd[p + "qkv_getter.1.weight"], sd[p + "qkv_getter.1.bias"]).transpose(1, 2)
    q, k, v = torch.split(qkv, C, dim=2)
    attn = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / (C ** 0.5), dim=2)
    o = torch.bmm(attn, v).transpose(1, 2)
    x = x + F.conv1d(o, sd[p + "project_out.weight"], sd[p + "project_out.bias"])
    return resnet_block(sd, p + "res2.", x, emb, groups)


def nearest_index_map(l_in: int, l_out: int) -> torch.Tensor:
    """Source index of F.interpolate(mode='nearest', size=l_out) (TrajUNet.py:93).

    ATen computes ``min(floor(dst * float(l_in / l_out)), l_in - 1)`` with an fp32 scale.
    """
    scale = torch.tensor(l_in / l_out, dtype=torch.float32)
    idx = torch.floor(torch.arange(l_out, dtype=torch.float32) * scale).to(torch.long)
    return idx.clamp_(max=l_in - 1)


def trajunet_forward(sd: SD, x: torch.Tensor, time: torch.Tensor, cat_attr: torch.Tensor, num_attr: torch.Tensor,
                     channel_schedule: List[int], diffusion_steps: int, res_blocks: int = 2,
                     taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """TrajUNet.forward, Models/TrajUNet.py:98-112.

    ``taps`` (optional dict) receives named intermediates for block-level parity tests.
    """
    stages = len(channel_schedule) - 1
    emb = embed_forward(sd, time, cat_attr, num_attr, diffusion_steps)
    if taps is not None:
        taps["embed"] = emb
    x = F.conv1d(x, sd["stem.weight"], sd["stem.bias"], padding=1)  # TrajUNet.py:21,107
    if taps is not None:
        taps["stem"] = x
    downs = [x]
    for i in range(stages):  # TrajUNet.py:65-78
        for j in range(res_blocks):
            x = resnet_block(sd, f"down_blocks.{i}.{j}.", x, emb)
        x = F.conv1d(x, sd[f"down_blocks.{i}.{res_blocks}.weight"], sd[f"down_blocks.{i}.{res_blocks}.bias"],
                     stride=2, padding=1)
        downs.append(x)
        if taps is not None:
            taps[f"down{i}"] = x
    x = attn_block(sd, "mid_attn_block.", downs[-1], emb)  # TrajUNet.py:110
    if taps is not None:
        taps["mid"] = x
    for i in range(stages):  # TrajUNet.py:81-95; up_stage[:-2] => only the first res_blocks-1 blocks run
        x = torch.cat([x, downs[-i - 1]], dim=1)
        for j in range(res_blocks - 1):
            x = resnet_block(sd, f"up_blocks.{i}.{j}.", x, emb)
        l_out = downs[-i - 2].shape[-1]
        x = x[:, :, nearest_index_map(x.shape[-1], l_out)]  # == F.interpolate(nearest, size=l_out)
        x = F.conv1d(x, sd[f"up_blocks.{i}.{res_blocks}.weight"], sd[f"up_blocks.{i}.{res_blocks}.bias"], padding=1)
        if taps is not None:
            taps[f"up{i}"] = x
    x = F.group_norm(x, 32, sd["head.0.weight"], sd["head.0.bias"], eps=1e-5)  # TrajUNet.py:43 default eps
    return F.conv1d(x, sd["head.1.weight"], sd["head.1.bias"], padding=1)
