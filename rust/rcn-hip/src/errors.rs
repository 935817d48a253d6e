//! rcn/src/errors.rs:1-13 -- the one error type of the public API.
#[derive(Debug, Clone)]
pub struct InvalidGrayscaleImageError;

impl std::error::Error for InvalidGrayscaleImageError {}

impl std::fmt::Display for InvalidGrayscaleImageError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str("InvalidGrayscaleImageError: Image provided was not Luma8 (grayscaled image)")
    }
}
