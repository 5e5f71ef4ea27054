"""Mutation fuzz of the asset loader (csrc/hrt_assets.cpp) against its Python restatement (oracle/orc_assets.py): a valid courtyard
(OBJ + MTL + three TGAs) and a PNG-textured triangle, one file at a time damaged by seeded byte flips, insertions, deletions,
truncations and token swaps.  Both sides must either fail (FormatException / EndOfStream class of errors) or succeed with
byte-identical meshes, materials and texels -- never crash, hang, or disagree.  No GPU."""
import os
import random
import shutil

import numpy as np
import pytest

from ilgpu_raytracing_amd import engine
from oracle import orc_assets as OA
from tests import asset_kit as K
from tests.test_assets import assert_mesh_equal

TEXT_TOKENS = [b"v", b"vt", b"vn", b"f", b"usemtl", b"mtllib", b"newmtl", b"map_Kd", b"map_d", b"Kd", b"d", b"Ni", b"illum", b"-1", b"0", b"1e39", b"nan",
               b"//", b"/", b" ", b"\t", b"\r\n", b"\n", b"#", b"1/2/3", b"-", b".", b"e", b"99999999999"]


def _mutate(data, rng, text):
    b = bytearray(data)
    if not b:
        return bytes(b)
    for _ in range(rng.choice([1, 1, 2, 4])):
        if not b:
            break
        op = rng.randrange(6 if text else 5)
        i = rng.randrange(len(b))
        if op == 0 and b:
            b[i] ^= 1 << rng.randrange(8)
        elif op == 1 and b:
            b[i] = rng.randrange(256) if not text else rng.choice(b"0123456789-+.e/ \n\tvfn#")
        elif op == 2:
            del b[i:i + rng.choice([1, 2, 7, 40])]
        elif op == 3:
            ins = bytes(rng.randrange(256) for _ in range(rng.choice([1, 3, 9]))) if not text else rng.choice(TEXT_TOKENS)
            b[i:i] = ins
        elif op == 4:
            del b[max(1, len(b) - rng.choice([1, 5, 64, len(b) // 2])):]
        else:
            j = rng.randrange(len(b))
            tok = rng.choice(TEXT_TOKENS)
            b[j:j + len(tok)] = tok
    return bytes(b)


def _agree(path):
    """-> 'ok' | 'format' ; raises AssertionError on any disagreement"""
    try:
        ref = OA.load_obj(path, 1.0, True)
    except OA.FormatError:
        ref = None
    except FileNotFoundError:
        ref = FileNotFoundError
    try:
        got = engine.load_obj(path, 1.0, True)
    except engine.AssetFormatError:
        got = None
    except FileNotFoundError:
        got = FileNotFoundError
    if ref is None or ref is FileNotFoundError or got is None or got is FileNotFoundError:
        assert (ref is None) == (got is None) and (ref is FileNotFoundError) == (got is FileNotFoundError), \
            "one side loads what the other rejects: oracle %s, product %s" % ("rejects" if ref is None else "loads/other", "rejects" if got is None else "loads/other")
        return "format"
    assert_mesh_equal(got, ref)
    return "ok"


@pytest.fixture(scope="module")
def pristine(tmp_path_factory):
    d = tmp_path_factory.mktemp("fuzz_src")
    obj = K.write_courtyard(str(d), grid=4)
    png_dir = os.path.join(str(d), "png")
    os.makedirs(png_dir)
    rng = np.random.default_rng(3)
    K.write_png(os.path.join(png_dir, "t.png"), rng.integers(0, 256, (6, 5, 4), dtype=np.uint8), ctype=6, interlace=True)
    with open(os.path.join(png_dir, "p.mtl"), "w") as f:
        f.write("newmtl a\nKd 1 1 1\nmap_Kd t.png\n")
    with open(os.path.join(png_dir, "p.obj"), "w") as f:
        f.write("mtllib p.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl a\nf 1/1 2/2 3/3\n")
    return str(d), obj, os.path.join(png_dir, "p.obj")


TARGETS = ["courtyard.obj", "courtyard.mtl", "textures/wall_diff.tga", "textures/leaf_mask.tga", "textures/floor.tga", "png/t.png", "png/p.mtl"]


@pytest.mark.parametrize("target", TARGETS)
def test_damaged_assets_load_or_fail_alike(pristine, tmp_path, target):
    src, obj, pobj = pristine
    work = str(tmp_path / "w")
    shutil.copytree(src, work)
    assert _agree(os.path.join(work, "courtyard.obj")) == "ok" and _agree(os.path.join(work, "png", "p.obj")) == "ok"
    entry = os.path.join(work, "png", "p.obj") if target.startswith("png/") else os.path.join(work, "courtyard.obj")
    victim = os.path.join(work, target)
    original = open(victim, "rb").read()
    text = target.endswith((".obj", ".mtl"))
    rng = random.Random(sum(target.encode()) * 7919)
    outcomes = {"ok": 0, "format": 0}
    rounds = int(os.environ.get("HRT_FUZZ_ROUNDS", "0")) or (120 if text else 160)
    for k in range(rounds):
        damaged = _mutate(original, rng, text)
        with open(victim, "wb") as f:
            f.write(damaged)
        try:
            outcomes[_agree(entry)] += 1
        except AssertionError as e:
            keep = os.path.join(str(tmp_path), "failing_case_%d" % k)
            shutil.copyfile(victim, keep)
            raise AssertionError("mutation %d of %s (kept at %s): %s" % (k, target, keep, e))
    assert outcomes["ok"] + outcomes["format"] == rounds and outcomes["format"] > 0       # the damage does reach the parsers
