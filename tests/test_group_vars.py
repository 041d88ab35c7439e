"""Group variables of the assigned task and of the upload file name:
scheduler_impl.rs:155-200 and storage.rs:150-215.  Oracle and product against the vectors transcribed from the
reference's own tests, and against each other on adversarial templates (values that contain other variables:
the reference's chained `str::replace` rescans the result of every pass)."""
import ctypes as C
import json
import os
import random

import pytest

from oracle import oracle_ffi as orc
from protocol_amd import engine as E
from protocol_amd import host

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "group_vars_kats.json")))


@pytest.mark.parametrize("k", KATS["group_vars"], ids=lambda k: k["name"])
def test_group_vars_kats(k):
    args = (k["in"], k["index"], k["size"], k["p2p"], k["gid"], k["count"])
    assert orc.group_vars(*args) == k["out"]
    assert host.group_vars(*args) == k["out"]


@pytest.mark.parametrize("k", KATS["upload_name"], ids=lambda k: k["name"])
def test_upload_name_kats(k):
    args = (k["in"], k["gid"], k["size"], k["index"], k["count"])
    assert orc.upload_name_vars(*args) == k["out"]
    assert host.upload_name_vars(*args) == k["out"]


@pytest.mark.parametrize("text,want", [
    ("", 0), ("+", 0), ("0", 0), ("1", 0), ("2", 1), ("+5", 4), ("05", 4), (" 5", 0), ("5 ", 0), ("-1", 0),
    ("abc", 0), ("1e3", 0), ("4294967295", 4294967294), ("4294967296", 0), ("99999999999999999999", 0),
])
def test_last_file_idx_follows_rust_u32_parse(text, want):
    """`parse::<u32>().unwrap_or(0).saturating_sub(1)` (scheduler_impl.rs:155-158)"""
    assert orc.last_file_idx(text) == want
    assert host.last_file_idx(text) == want


def test_chained_replace_semantics_product_equals_oracle():
    """Values may themselves contain variables; a later pass of the chain sees them, an earlier one does not."""
    rng = random.Random(7)
    names = ["${GROUP_INDEX}", "${GROUP_SIZE}", "${NEXT_P2P_ADDRESS}", "${GROUP_ID}", "${TOTAL_UPLOAD_COUNT}",
             "${LAST_FILE_IDX}", "${NODE_GROUP_ID}", "${NODE_GROUP_SIZE}", "${NODE_GROUP_INDEX}",
             "${TOTAL_UPLOAD_COUNT_AFTER}", "${CURRENT_FILE_INDEX}", "${", "}", "$", "{GROUP_ID}", "x", "/", "-", "ü"]
    for _ in range(400):
        text = "".join(rng.choice(names) for _ in range(rng.randint(0, 12)))
        p2p = "".join(rng.choice(names) for _ in range(rng.randint(0, 3)))
        gid = "".join(rng.choice(names[3:]) for _ in range(rng.randint(0, 3)))
        count = rng.choice(["0", "1", "7", "+3", "x", "", "${LAST_FILE_IDX}"])
        idx, size = rng.randint(0, 70), rng.randint(0, 70)
        assert host.group_vars(text, idx, size, p2p, gid, count) == orc.group_vars(text, idx, size, p2p, gid, count)
        assert host.volume_vars(text, gid) == orc.volume_vars(text, gid)
        g = rng.choice([None, gid])
        n = rng.choice([0, 1, 2, 10 ** 12])
        assert host.upload_name_vars(text, g, size, idx, n) == orc.upload_name_vars(text, g, size, idx, n)
    # an id that contains a later variable is expanded by the later pass, one with an earlier variable is not
    assert host.group_vars("${GROUP_ID}", 3, 9, "", "a${TOTAL_UPLOAD_COUNT}b${GROUP_INDEX}", "5") == "a5b${GROUP_INDEX}"


def test_buffer_protocol():
    L = E.lib()
    v = E.GroupVars(1, 2, b"", b"gid", b"3")
    need = C.c_size_t(0)
    assert L.pm_host_group_vars(b"a${GROUP_ID}b", C.byref(v), None, 0, C.byref(need)) == 0 and need.value == 6
    small = C.create_string_buffer(5)
    assert L.pm_host_group_vars(b"a${GROUP_ID}b", C.byref(v), small, 5, C.byref(need)) == E.PM_ERANGE
    ok = C.create_string_buffer(6)
    assert L.pm_host_group_vars(b"a${GROUP_ID}b", C.byref(v), ok, 6, C.byref(need)) == 0 and ok.value == b"agidb"
    assert L.pm_host_group_vars(None, C.byref(v), ok, 6, C.byref(need)) == E.PM_EINVAL
    assert L.pm_host_volume_vars(b"/data/${GROUP_ID}", None, ok, 6, C.byref(need)) == E.PM_EINVAL
