"""rust/rcn-hip is the drop-in `rcn` crate a maintainer of the reference would build against librcn_hip.so.  This image has no
cargo / rustc, so the source cannot be compiled here; what CAN be held is that its public surface is the reference's: every
public item below carries exactly the reference's signature (rcn/src/rcn.rs:58-75, 82, 126-133;
rcn/src/utils/kernel.rs:61-100, 219-236; rcn/src/lib.rs:27), so rcn/src/main.rs, rcn/benches/*.rs and backend/src/main.rs
compile against it unchanged.  When /root/reference is present (this container, not the GPU box) the table itself is checked
against the reference's source text."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "rcn-hip", "src")
REF = "/root/reference/rcn/src"


def _norm(s: str) -> str:
    s = re.sub(r"//[^\n]*", "", s)
    s = re.sub(r"\s+", " ", s).strip()
    s = re.sub(r"\s*([(),:<>&;=+{}\[\]])\s*", r"\1", s)     # spacing around punctuation is not part of a signature
    return s.replace(",)", ")").replace(",>", ">")            # nor is a trailing comma


def _fn_signature(text: str, name: str, after: str = "") -> str:
    """`fn name ...` up to the body's `{` or the declaration's `;`, searched after the first occurrence of `after`."""
    start = text.index(after) if after else 0
    m = re.search(r"(pub\s+)?fn\s+" + re.escape(name) + r"\b", text[start:])
    assert m, f"fn {name} not found"
    i = start + m.start()
    depth, j = 0, i
    while j < len(text):
        ch = text[j]
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            if not (ch == ">" and text[j - 1] == "-"):        # `->` is not a closing bracket
                depth -= 1
        elif ch in "{;" and depth == 0:
            break
        j += 1
    return _norm(text[i:j])


def _item(text: str, head: str) -> str:
    """a struct / enum / trait header: from `head` to its closing `;` or matching `}` (where-clauses included)"""
    i = text.index(head)
    j = i
    depth = sq = 0
    while j < len(text):
        if text[j] == "[":
            sq += 1                                          # `;` separates the rows inside matrix![...]
        elif text[j] == "]":
            sq -= 1
        elif text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return _norm(text[i:j + 1])
        elif text[j] == ";" and depth == 0 and sq == 0:
            return _norm(text[i:j + 1])
        j += 1
    raise AssertionError(head)


# the reference's public surface for this path; file -> [(kind, key, anchor, signature as the reference spells it)]
SURFACE = {
    "rcn.rs": [
        ("fn", "new", "impl<'a> RCN<'a>",
         "pub fn new(classes: usize, convpool_cfg: Vec<RCNLayer>, feedforward_cfg: Vec<usize>, training_path: &'a str, testing_path: &'a str) -> Self"),
        ("fn", "classify", "impl<'a> RCN<'a>",
         "pub fn classify(&self, img_path: &str) -> Result<usize, Box<dyn std::error::Error>>"),
        ("fn", "train", "impl<'a> RCN<'a>",
         "pub fn train(&mut self, batch_size: usize, epochs: usize, eta: f64, training_class_size_limit: usize, testing_class_size_limit: usize) -> Result<(), ImageError>"),
        ("item", "pub struct Weights", "", "pub struct Weights(pub DMatrix<f64>);"),
        ("item", "pub struct Bias", "", "pub struct Bias(pub DVector<f64>);"),
        ("item", "pub enum RCNLayer", "", "pub enum RCNLayer { Convolve2D(Padding), Pool2D(Pooling), }"),
    ],
    "utils/kernel.rs": [
        ("fn", "convolve_2d", "pub trait Convolve2D",
         "fn convolve_2d<R2, C2, S2>(&self, kernel: &Matrix<N, R2, C2, S2>, padding: &Padding) -> DMatrix<N> where R2: Dim, C2: Dim, S2: Storage<N, R2, C2>"),
        ("fn", "convolve_2d_separated", "pub trait Convolve2D", "fn convolve_2d_separated(&self, op: SeparableOperator, padding: &Padding) -> DMatrix<N>"),
        ("fn", "relu", "pub trait Convolve2D", "fn relu(&self) -> DMatrix<N>"),
        ("fn", "pool_2d", "pub trait Pool2D", "fn pool_2d(&self, padding: &Padding, pooling: &Pooling) -> DMatrix<N>"),
        ("head", "pub trait Convolve2D", "",
         "pub trait Convolve2D<N, R1, C1, S1> where N: Scalar + Zero + One + AddAssign + Sub<Output = N> + Mul<Output = N> + Copy + PartialOrd, R1: Dim, C1: Dim, S1: Storage<N, R1, C1>,"),
        ("head", "pub trait Pool2D", "",
         "pub trait Pool2D<N, R, C, S> where N: Scalar + Zero + One + AddAssign + Sub<Output = N> + Mul<Output = N> + Copy + PartialOrd, R: Dim, C: Dim, S: Storage<N, R, C>,"),
        ("item", "pub enum SeparableOperator", "", "pub enum SeparableOperator { Top, Bottom, Left, Right, }"),
        ("item", "pub enum Padding", "", "pub enum Padding { None, Same, }"),
        ("item", "pub enum Pooling", "", "pub enum Pooling { Average, Max, }"),
        ("item", "pub const __TOP_SOBEL", "", "pub const __TOP_SOBEL: Matrix3<f64> = matrix![1.0, 2.0, 1.0; 0.0, 0.0, 0.0; -1.0, -2.0, -1.0];"),
    ],
    "lib.rs": [
        ("fn", "get_pixel_matrix", "", "pub fn get_pixel_matrix(image: &DynamicImage) -> Result<DMatrix<f64>, InvalidGrayscaleImageError>"),
    ],
}


def _extract(text, kind, key, anchor):
    if kind == "fn":
        return _fn_signature(text, key, anchor)
    if kind == "head":                         # trait header up to its opening brace
        i = text.index(key)
        return _norm(text[i:text.index("{", i)])
    return _item(text, key)


@pytest.mark.parametrize("fname", sorted(SURFACE))
def test_dropin_crate_has_the_reference_public_signatures(fname):
    text = open(os.path.join(CRATE, fname)).read()
    for kind, key, anchor, want in SURFACE[fname]:
        assert _extract(text, kind, key, anchor) == _norm(want), (fname, key)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("fname", sorted(SURFACE))
def test_signature_table_is_the_reference_text(fname):
    text = open(os.path.join(REF, fname)).read()
    for kind, key, anchor, want in SURFACE[fname]:
        assert _extract(text, kind, key, anchor) == _norm(want), (fname, key)


def test_dropin_crate_layout_and_bincode_field_order():
    cargo = open(os.path.join(ROOT, "rust", "rcn-hip", "Cargo.toml")).read()
    assert re.search(r'^name = "rcn"$', cargo, re.M) and "rcn-hip-sys" in cargo          # same crate name: `use rcn::rcn::RCN` keeps working
    lib = open(os.path.join(CRATE, "lib.rs")).read()
    assert "pub mod rcn;" in lib and "pub mod utils;" in lib                                # rcn::rcn::*, rcn::utils::kernel::*
    assert "pub mod kernel;" in open(os.path.join(CRATE, "utils", "mod.rs")).read()
    # bincode writes fields in declaration order (rcn.rs:15-25): the model fields come first and in the reference's order; the device
    # handle is skipped by serde
    rcn = open(os.path.join(CRATE, "rcn.rs")).read()
    body = rcn[rcn.index("pub struct RCN<'a> {"):]
    body = body[:body.index("\n}")]
    fields = re.findall(r"^\s*(?:#\[serde\(skip\)\]\s*)?([a-z_]+):", body, re.M)
    assert fields == ["classes", "convpool_cfg", "feedforward_cfg", "layer_weights", "layer_bias", "scale_set", "training_path", "testing_path", "device"]
    assert re.search(r"#\[serde\(skip\)\]\s*device:", body)
    # every arithmetic seam goes through the C ABI: no CPU arithmetic fallback in the crate
    for seam in ("load_data", "train_set_epoch", "evaluate_set", "classify_image", "init_params"):
        assert f"ctx.{seam}(" in rcn, seam
    ker = open(os.path.join(CRATE, "utils", "kernel.rs")).read()
    for call in ("ctx.convolve_2d(", "ctx.convolve_2d_separated(", "ctx.relu(", "ctx.pool_2d("):
        assert call in ker, call
    # and every wrapper method the crate calls exists in rcn-hip-sys's safe Context
    sys_rs = open(os.path.join(ROOT, "rust", "rcn-hip-sys", "src", "lib.rs")).read()
    for m in set(re.findall(r"ctx\.([a-z_0-9]+)\(", rcn + ker)):
        assert re.search(r"pub fn " + m + r"\b", sys_rs), f"Context::{m} is not defined in rcn-hip-sys"
    for cst in set(re.findall(r"sys::(RCN_HIP_[A-Z0-9_]+)", rcn + ker)):
        assert re.search(r"pub const " + cst + r"\b", sys_rs), cst
