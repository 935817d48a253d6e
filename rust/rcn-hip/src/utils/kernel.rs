//! The operator traits of rcn/src/utils/kernel.rs (`Convolve2D` :61-100, `Pool2D` :219-236) with the reference's method
//! signatures, implemented by forwarding to the HIP operator API (`rcn_hip_convolve_2d`, `rcn_hip_convolve_2d_separated`,
//! `rcn_hip_relu`, `rcn_hip_pool_2d`: f64 on the device in the reference's summation order, bit-identical results).
//!
//! One deliberate narrowing: the reference blanket-implements the traits for every scalar `N`; the device arithmetic is f64, the
//! only scalar the reference crate itself instantiates (rcn.rs:28,31,49; benches/convolve.rs), so the impls here are for
//! `Matrix<f64, R, C, S>`.  The trait declarations keep the generic form, so code written against them compiles unchanged.
use nalgebra::{matrix, DMatrix, Dim, Matrix, Matrix3, Scalar, Storage};
use num::{One, Zero};
use rcn_hip_sys as sys;
use std::ops::{AddAssign, Mul, Sub};
use serde::{Deserialize, Serialize};
use std::cell::RefCell;

/// utils/kernel.rs:16-21
#[derive(Clone, Copy)]
pub enum SeparableOperator {
    Top,
    Bottom,
    Left,
    Right,
}

/// utils/kernel.rs:25-28 (bincode tag = variant index = `rcn_hip_padding`)
#[derive(Serialize, Deserialize)]
pub enum Padding {
    None,
    Same,
}

/// utils/kernel.rs:32-35 (bincode tag = variant index = `rcn_hip_pooling`)
#[derive(Serialize, Deserialize)]
pub enum Pooling {
    Average,
    Max,
}

// utils/kernel.rs:56-59: the full 3x3 operators the separated pairs multiply out to
pub const __TOP_SOBEL: Matrix3<f64> = matrix![1.0, 2.0, 1.0; 0.0, 0.0, 0.0; -1.0, -2.0, -1.0];
pub const __BOTTOM_SOBEL: Matrix3<f64> = matrix![-1.0, -2.0, -1.0; 0.0, 0.0, 0.0; 1.0, 2.0, 1.0];
pub const __LEFT_SOBEL: Matrix3<f64> = matrix![1.0, 0.0, -1.0; 2.0, 0.0, -2.0; 1.0, 0.0, -1.0];
pub const __RIGHT_SOBEL: Matrix3<f64> = matrix![-1.0, 0.0, 1.0; -2.0, 0.0, 2.0; -1.0, 0.0, 1.0];

pub(crate) fn padding_tag(p: &Padding) -> i32 {
    match p {
        Padding::None => sys::RCN_HIP_PAD_NONE,
        Padding::Same => sys::RCN_HIP_PAD_SAME,
    }
}

pub(crate) fn pooling_tag(p: &Pooling) -> i32 {
    match p {
        Pooling::Average => sys::RCN_HIP_POOL_AVERAGE,
        Pooling::Max => sys::RCN_HIP_POOL_MAX,
    }
}

fn op_tag(op: SeparableOperator) -> i32 {
    match op {
        SeparableOperator::Top => 0,
        SeparableOperator::Bottom => 1,
        SeparableOperator::Left => 2,
        SeparableOperator::Right => 3,
    }
}

thread_local! {
    // the operator calls are stateless; one small context per thread serves them (a context is Send, not Sync)
    static OPS: RefCell<Option<sys::Context>> = RefCell::new(None);
}

fn with_ops<T>(f: impl FnOnce(&mut sys::Context) -> T) -> T {
    OPS.with(|slot| {
        let mut slot = slot.borrow_mut();
        let ctx = slot.get_or_insert_with(|| {
            let layers = [sys::rcn_hip_layer { kind: sys::RCN_HIP_LAYER_CONVOLVE2D, arg: sys::RCN_HIP_PAD_SAME }];
            sys::Context::new(1, &layers, &[1], 3, 3, sys::RCN_HIP_F64, 0).expect("rcn_hip_create (operator context)")
        });
        f(ctx)
    })
}

/// Column-major copy of any f64 matrix view (nalgebra's own storage order; what the C ABI takes).
fn colmajor<R: Dim, C: Dim, S: Storage<f64, R, C>>(m: &Matrix<f64, R, C, S>) -> Vec<f64> {
    m.iter().copied().collect()
}

pub trait Convolve2D<N, R1, C1, S1>
where
    N: Scalar + Zero + One + AddAssign + Sub<Output = N> + Mul<Output = N> + Copy + PartialOrd,
    R1: Dim,
    C1: Dim,
    S1: Storage<N, R1, C1>,
{
    /// Cross-correlation with `kernel` under `Padding::None` / `Padding::Same` (utils/kernel.rs:110-194, including the
    /// pad-copy index behaviour of :154-158).  Panics where the reference panics (:123-135).
    fn convolve_2d<R2, C2, S2>(
        &self,
        kernel: &Matrix<N, R2, C2, S2>,
        padding: &Padding,
    ) -> DMatrix<N>
    where
        R2: Dim,
        C2: Dim,
        S2: Storage<N, R2, C2>;

    /// relu(conv(conv(self, column 3x1), row 1x3)) with the Sobel pair of `op` (utils/kernel.rs:196-207).
    fn convolve_2d_separated(&self, op: SeparableOperator, padding: &Padding) -> DMatrix<N>;

    /// Activate the convoluted matrix with ReLU (utils/kernel.rs:209-216).
    fn relu(&self) -> DMatrix<N>;
}

impl<R1, C1, S1> Convolve2D<f64, R1, C1, S1> for Matrix<f64, R1, C1, S1>
where
    R1: Dim,
    C1: Dim,
    S1: Storage<f64, R1, C1>,
{
    fn convolve_2d<R2, C2, S2>(
        &self,
        kernel: &Matrix<f64, R2, C2, S2>,
        padding: &Padding,
    ) -> DMatrix<f64>
    where
        R2: Dim,
        C2: Dim,
        S2: Storage<f64, R2, C2>,
    {
        let (r, c) = self.shape();
        let (kr, kc) = kernel.shape();
        let (m, k) = (colmajor(self), colmajor(kernel));
        with_ops(|ctx| {
            let (or, oc) = ctx.conv_out_shape(r, c, kr, kc, padding_tag(padding));      // panics like kernel.rs:123-135
            let mut out = vec![0.0; or * oc];
            ctx.convolve_2d(&m, r, c, &k, kr, kc, padding_tag(padding), &mut out).expect("rcn_hip_convolve_2d");
            DMatrix::from_vec(or, oc, out)
        })
    }

    fn convolve_2d_separated(&self, op: SeparableOperator, padding: &Padding) -> DMatrix<f64> {
        let (r, c) = self.shape();
        let m = colmajor(self);
        with_ops(|ctx| {
            let (or, oc) = match padding {
                Padding::Same => (r, c),
                Padding::None => (r.saturating_sub(2), c.saturating_sub(2)),
            };
            let mut out = vec![0.0; or * oc];
            ctx.convolve_2d_separated(&m, r, c, op_tag(op), padding_tag(padding), &mut out).expect("rcn_hip_convolve_2d_separated");
            DMatrix::from_vec(or, oc, out)
        })
    }

    fn relu(&self) -> DMatrix<f64> {
        let (r, c) = self.shape();
        let m = colmajor(self);
        with_ops(|ctx| {
            let mut out = vec![0.0; r * c];
            ctx.relu(&m, &mut out).expect("rcn_hip_relu");
            DMatrix::from_vec(r, c, out)
        })
    }
}

pub trait Pool2D<N, R, C, S>
where
    N: Scalar + Zero + One + AddAssign + Sub<Output = N> + Mul<Output = N> + Copy + PartialOrd,
    R: Dim,
    C: Dim,
    S: Storage<N, R, C>,
{
    /// 2x2 / stride-2 pooling; `Padding::Same` zero-pads odd dimensions at the bottom / right, `Padding::None` truncates
    /// (utils/kernel.rs:245-349).  `Pooling::Average` panics "Not implemented" as in the reference (:283-285).
    fn pool_2d(&self, padding: &Padding, pooling: &Pooling) -> DMatrix<N>;
}

impl<R, C, S> Pool2D<f64, R, C, S> for Matrix<f64, R, C, S>
where
    R: Dim,
    C: Dim,
    S: Storage<f64, R, C>,
{
    fn pool_2d(&self, padding: &Padding, pooling: &Pooling) -> DMatrix<f64> {
        let (r, c) = self.shape();
        let m = colmajor(self);
        with_ops(|ctx| {
            let (or, oc) = ctx.pool_out_shape(r, c, padding_tag(padding));               // panics like kernel.rs:246-251
            let mut out = vec![0.0; or * oc];
            ctx.pool_2d(&m, r, c, padding_tag(padding), pooling_tag(pooling), &mut out).expect("rcn_hip_pool_2d");
            DMatrix::from_vec(or, oc, out)
        })
    }
}
