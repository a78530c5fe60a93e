"""Pure-Python restatement of the reference's ETL step, DataPreprocess/preprocess.cpp:50-155 (test oracle; see oracle/__init__.py).

PARITY UNPINNED: the reference ships no input/output pair for this step and cannot be built here (it links libtorch).  Written by
reading the source; the C++ product tool (trajectorydiffusion_amd/DataPreprocess/preprocess.cpp) is compared against it bit for bit.

Where the reference leaves the result open (torch::argsort without stable=True on equal time stamps; the racy append order of its
OpenMP loop, :141-149) this restatement fixes: stable sort, orders in order of first appearance.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Tuple

import torch

_LONG = re.compile(r"[ \t\n\v\f\r]*([+-]?\d+)")
_DBL = re.compile(r"[ \t\n\v\f\r]*([+-]?(?:inf(?:inity)?|nan(?:\([0-9a-z_]*\))?|0x(?:[0-9a-f]+\.?[0-9a-f]*|\.[0-9a-f]+)(?:p[+-]?\d+)?|"
                  r"(?:\d+\.?\d*|\.\d+)(?:e[+-]?\d+)?))", re.IGNORECASE)


def stol(tok: str) -> Optional[int]:
    """std::stol (:58): None where it throws (no conversion, or out of the range of a 64-bit long)."""
    m = _LONG.match(tok)
    if not m:
        return None
    v = int(m.group(1))
    return v if -2 ** 63 <= v < 2 ** 63 else None


def stod(tok: str) -> Optional[float]:
    """std::stod (:60,62): None where it throws (no conversion, or strtod reports ERANGE: overflow to infinity / underflow)."""
    m = _DBL.match(tok)
    if not m:
        return None
    txt = m.group(1)
    low = txt.lower().lstrip("+-")
    if low.startswith("inf"):
        return -math.inf if txt.startswith("-") else math.inf
    if low.startswith("nan"):
        return math.nan   # (sign / payload of a NaN are not compared)
    if low.startswith("0x"):
        try:
            v = float.fromhex(txt)
        except OverflowError:
            return None
    else:
        v = float(txt)
    if math.isinf(v):
        return None                                    # overflow: ERANGE
    digits = re.sub(r"(?i)(e|p)[+-]?\d+$", "", low)
    if re.search(r"[1-9a-f]", digits.replace("0x", "")) and (v == 0.0 or abs(v) < 2.2250738585072014e-308):
        return None                                    # underflow: ERANGE
    return v


def parse_line(line: str):
    """parseLine (:50-65) with std::getline's end-of-line behaviour: once the end of the line was reached inside a field, later
    getline calls fail and leave their string untouched (`token` keeps its previous text); after a trailing comma the next field
    is ""."""
    fields = line.split(",")
    driver = fields[0] if len(fields) > 0 else ""
    order = fields[1] if len(fields) > 1 else ""
    token = ""
    vals = []
    for k, conv in ((2, stol), (3, stod), (4, stod)):
        if k < len(fields):
            token = fields[k]
        v = conv(token)
        if v is None:
            return None
        vals.append(v)
    return driver, order, vals[0], vals[1], vals[2]


def parse_file(text: str) -> List[Tuple[str, str, torch.Tensor, torch.Tensor]]:
    """parseFile (:96-155): list of (order_id, driver_id, float64 (N, 2) {lon, lat}, int64 (N) time), each sorted by time."""
    orders: Dict[str, Tuple[str, List[Tuple[int, float, float]]]] = {}
    for line in text.split("\n")[:-1] if text.endswith("\n") else text.split("\n"):
        info = parse_line(line)
        if info is None:
            continue                                   # :117-120
        driver, order, t, lon, lat = info
        if order in orders:
            orders[order][1].append((t, lon, lat))     # :123-124
        else:
            orders[order] = (driver, [(t, lon, lat)])  # :126-128
    out = []
    for order, (driver, pts) in orders.items():        # dict = insertion order
        pts = sorted(pts, key=lambda p: p[0])          # stable
        out.append((order, driver, torch.tensor([[p[1], p[2]] for p in pts], dtype=torch.float64).reshape(-1, 2),
                    torch.tensor([p[0] for p in pts], dtype=torch.int64)))
    return out
