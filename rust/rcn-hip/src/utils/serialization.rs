//! Checkpoint layout of `Weights` / `Bias` (reference: rcn/src/utils/serialization.rs:11-151): `Weights` is the struct
//! `{ dims: (usize, usize), data: Vec<f64> }` with `data` column-major, `Bias` a plain sequence of f64.  bincode does not
//! encode field or type names, so mirror types with the same field order produce and accept the reference's bytes
//! (mercer_research_amd/checkpoint.py and csrc/host/formats.hpp read and write the same layout, byte for byte).
use crate::rcn::{Bias, Weights};
use nalgebra::{DMatrix, DVector};
use serde::{Deserialize, Deserializer, Serialize, Serializer};

#[derive(Serialize, Deserialize)]
#[serde(rename = "Weights")]
struct WeightsRepr {
    dims: (usize, usize),
    data: Vec<f64>,
}

impl Serialize for Weights {
    fn serialize<S: Serializer>(&self, serializer: S) -> Result<S::Ok, S::Error> {
        WeightsRepr { dims: self.0.shape(), data: self.0.as_slice().to_vec() }.serialize(serializer)
    }
}

impl<'de> Deserialize<'de> for Weights {
    fn deserialize<D: Deserializer<'de>>(deserializer: D) -> Result<Self, D::Error> {
        let r = WeightsRepr::deserialize(deserializer)?;
        if r.data.len() != r.dims.0 * r.dims.1 {
            return Err(serde::de::Error::custom("Weights: data length does not match dims"));
        }
        Ok(Weights(DMatrix::from_vec(r.dims.0, r.dims.1, r.data)))
    }
}

impl Serialize for Bias {
    fn serialize<S: Serializer>(&self, serializer: S) -> Result<S::Ok, S::Error> {
        serializer.collect_seq(self.0.iter())
    }
}

impl<'de> Deserialize<'de> for Bias {
    fn deserialize<D: Deserializer<'de>>(deserializer: D) -> Result<Self, D::Error> {
        Ok(Bias(DVector::from_vec(Vec::<f64>::deserialize(deserializer)?)))
    }
}
