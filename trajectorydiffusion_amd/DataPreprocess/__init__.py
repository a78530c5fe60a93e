"""Host side of the ETL step (reference DataPreprocess/preprocess.cpp): run the C++ tool on one Didi GPS dump and save the result as
the `.pt` file DidiTrajectoryDataset loads (Dataset/DidiDataset.py: a list of (order_id, driver_id, float64 (N, 2), int64 (N))).

    from trajectorydiffusion_amd.DataPreprocess import preprocess_file
    preprocess_file("/data/xian/nov/gps_20161101")          # writes /data/xian/nov/gps_20161101.pt, like the reference (:153)
"""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile
from typing import List, Optional, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(_HERE, "preprocess")

Traj = Tuple[str, str, torch.Tensor, torch.Tensor]


def build_tool() -> str:
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return TOOL


def read_container(path: str) -> List[Traj]:
    """Parse the tool's binary container (layout in preprocess.cpp's header)."""
    buf = open(path, "rb").read()
    if buf[:8] != b"TDXTRAJ1":
        raise ValueError(f"{path}: not a TDXTRAJ1 container")
    (n,), pos = struct.unpack_from("<Q", buf, 8), 16
    out: List[Traj] = []
    for _ in range(n):
        ids = []
        for _ in range(2):
            (ln,) = struct.unpack_from("<I", buf, pos)
            ids.append(buf[pos + 4:pos + 4 + ln].decode("utf-8", errors="surrogateescape"))
            pos += 4 + ln
        (npts,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        times = np.frombuffer(buf, dtype="<i8", count=npts, offset=pos).copy()
        pos += 8 * npts
        lonlat = np.frombuffer(buf, dtype="<f8", count=2 * npts, offset=pos).reshape(npts, 2).copy()
        pos += 16 * npts
        out.append((ids[0], ids[1], torch.from_numpy(lonlat), torch.from_numpy(times)))
    if pos != len(buf):
        raise ValueError(f"{path}: {len(buf) - pos} trailing bytes")
    return out


def preprocess_file(csv_path: str, out_path: Optional[str] = None, threads: int = 8) -> str:
    """preprocess.cpp's parseFile (:96-155) for one file: returns the path of the written `.pt`."""
    if not os.path.exists(TOOL):
        build_tool()
    out_path = csv_path + ".pt" if out_path is None else out_path
    with tempfile.TemporaryDirectory() as tmp:
        binp = os.path.join(tmp, "traj.bin")
        subprocess.run([TOOL, csv_path, binp, str(threads)], check=True, stdout=subprocess.DEVNULL)
        torch.save(read_container(binp), out_path)
    return out_path


def preprocess_dir(data_dir: str, threads: int = 8) -> List[str]:
    """preprocess.cpp's main (:163-186): every file whose name starts with 'g' and does not end in 't'."""
    done = []
    for name in sorted(os.listdir(data_dir)):
        if name and name[0] == "g" and name[-1] != "t":
            done.append(preprocess_file(os.path.join(data_dir, name), threads=threads))
    return done
