//! `rcn` on an MI355X: the reference crate's public surface (rcn/src/lib.rs, rcn/src/rcn.rs, rcn/src/utils/kernel.rs) with every
//! arithmetic seam forwarded to `librcn_hip.so` through `rcn-hip-sys`.  Uncompiled in this repository (no Rust toolchain in the
//! image); `tests/test_rust_dropin.py` checks the public signatures below against the reference's.
pub mod rcn;
pub mod utils;

mod errors;

use errors::InvalidGrayscaleImageError;
use image::DynamicImage;
use nalgebra::DMatrix;

/// Luma8 / LumaA8 image -> `DMatrix<f64>` with rows = y, columns = x, values 0..255 (reference: rcn/src/lib.rs:27-41).
/// Kept for source compatibility (`rcn/benches/convolve.rs` calls it); the training and classification paths hand the
/// decoded bytes to the device directly and never build this matrix.
pub fn get_pixel_matrix(image: &DynamicImage) -> Result<DMatrix<f64>, InvalidGrayscaleImageError> {
    let (w, h, px) = gray_bytes(image).ok_or(InvalidGrayscaleImageError)?;
    Ok(DMatrix::from_row_iterator(h, w, px.into_iter().map(f64::from)))
}

/// (width, height, row-major luma bytes) of a grayscale image; alpha is ignored.  `None` for anything that is not Luma8 / LumaA8.
pub(crate) fn gray_bytes(image: &DynamicImage) -> Option<(usize, usize, Vec<u8>)> {
    match image {
        DynamicImage::ImageLuma8(g) => Some((g.width() as usize, g.height() as usize, g.as_raw().clone())),
        DynamicImage::ImageLumaA8(g) => Some((g.width() as usize, g.height() as usize, g.pixels().map(|p| p.0[0]).collect())),
        _ => None,
    }
}
