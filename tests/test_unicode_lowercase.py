"""`str::to_lowercase` in the model rule of GpuSpecs::meets (shared/src/models/node.rs:465, :470) is the full Unicode mapping,
not ASCII: the product (pm_host.cpp, pm::to_lowercase) and the oracle (pm_oracle.c, orc_to_lowercase) each implement it —
char::to_lowercase per code point (up to three for one), a capital sigma that ends a word as the final form, str::trim over
Unicode white space — and both are held here to CPython's `str.lower`, which implements the same algorithm over the same
tables' source (Unicode 13.0; tools/make_unicode_tables.py).  CPU-only (host helpers)."""
import random

import pytest

from oracle import oracle_ffi as orc
from protocol_amd import host

IMPLS = [("product", host.to_lowercase), ("oracle", orc.to_lowercase)]


@pytest.mark.parametrize("name,lower", IMPLS)
def test_every_code_point_lowercases_as_in_cpython(name, lower):
    cps = [c for c in range(1, 0x110000) if not 0xD800 <= c <= 0xDFFF]
    for i in range(0, len(cps), 8192):
        text = "".join(chr(c) for c in cps[i:i + 8192])
        assert lower(text) == text.lower(), (name, hex(cps[i]))


@pytest.mark.parametrize("name,lower", IMPLS)
def test_vectors(name, lower):
    for text, want in [("NVIDIA H100 80GB", "nvidia h100 80gb"), ("ÀÉÎÕÜ Ñ ß", "àéîõü ñ ß"), ("İstanbul", "i̇stanbul"),
                       ("ΣΑΣ", "σας"), ("ΑΣ.", "ας."), ("Σ", "σ"), ("ΑΣΑ", "ασα"), ("ΑΣ́", "ας́"), ("ΆΣ Β", "άς β"),
                       ("K80 Ω", "k80 ω"), ("ǅ Ǆ", "ǆ ǆ"), ("𐐀𐐁", "𐐨𐐩"), ("ＡＢＣ", "ａｂｃ"), ("", "")]:
        assert lower(text) == want == text.lower(), (name, text)


def test_sigma_rule_on_random_greek_words():
    rng = random.Random(5)
    alphabet = ["Σ", "Α", "σ", "a", "A", " ", ".", "́", "'", "­", "1", "ʰ", ":"]
    for _ in range(3000):
        text = "".join(rng.choice(alphabet) for _ in range(rng.randint(1, 12)))
        assert host.to_lowercase(text) == orc.to_lowercase(text) == text.lower(), repr(text)


@pytest.mark.parametrize("spec,req,want", [
    ("Ünïcode GPÜ 80GB", "ünïcode_gpü", True),          # an ASCII-only to_lowercase leaves Ü and Ï alone: no match
    ("K80", "k80", True),                           # the Kelvin sign lower-cases to 'k'
    ("NVIDIA H100 80GB HBM3", " h100 ", True), # str::trim strips NO-BREAK SPACE and EM SPACE
    ("ΤΕΣΛΑΣ V100", "τεσλας", True),                     # word-final sigma on both sides
    ("ΤΕΣΛΑΣ V100", "τεσλασ", False),                    # ... which a plain per-character mapping would get wrong
    ("İ100", "i100", False),                             # U+0130 becomes 'i' + COMBINING DOT ABOVE: "i̇100" does not contain "i100"
    ("İ100", "i̇100", True),
    ("a100", "A100,Ü", True),
])
def test_model_rule_with_non_ascii_strings(spec, req, want):
    assert host.model_matches(spec, req) == orc.model_matches(spec, req) == want
    # ... and the reference's rule, spelled with Python's own lower(): the four `contains`
    ns = spec.lower().replace(" ", "_")
    got = False
    for part in req.split(","):
        nr = part.strip().lower().replace(" ", "_")
        got = got or nr in ns or ns in nr or nr.replace("_", "") in ns.replace("_", "") or ns.replace("_", "") in nr.replace("_", "")
    assert got == want


def test_product_and_oracle_agree_on_random_model_strings():
    rng = random.Random(9)
    alphabet = list("aAbB19 _,-") + ["Ü", "ü", "Σ", "σ", "ς", "İ", " ", "ß", "K", "K", "Ⱥ", "ⱥ", "́"]
    for _ in range(4000):
        spec = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 14)))
        req = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 14)))
        assert host.model_matches(spec, req) == orc.model_matches(spec, req), (spec, req)
