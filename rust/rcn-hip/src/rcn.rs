//! `RCN` with the reference's public API (rcn/src/rcn.rs:13-75, 82, 126-133) and checkpoint layout, its four arithmetic seams --
//! `flatten_feature_set` (:317), `gen_scales` (:230), `train_batch` (:176), `classify_test` (:105) -- running on the device.
//!
//! What stays on the host is what the reference does on the host around those seams: directory listing, per-class sampling
//! without replacement, PNG decode, the shuffle, the epoch line.  Both data sets are uploaded ONCE (`rcn_hip_load_data`) and stay
//! in HBM; an epoch is one `rcn_hip_train_set_epoch` call (the shuffled order goes down as an index list) plus one
//! `rcn_hip_evaluate_set`.  Parameters live on the device during `train` and are pulled back into `layer_weights` / `layer_bias`
//! before it returns, so `bincode::serialize(&model)` (rcn/src/main.rs:77) writes the trained model.
use crate::utils::kernel::{padding_tag, pooling_tag, Padding, Pooling};
use image::{io::Reader as ImageReader, ImageError};
use nalgebra::{DMatrix, DVector};
use rand::seq::SliceRandom;
use rcn_hip_sys as sys;
use serde::{Deserialize, Serialize};
use std::sync::Mutex;
use std::{fs, path::PathBuf};

#[derive(Serialize, Deserialize)]
/// Rust Convolutional Neural Network (RCN) -- field order = bincode order (rcn.rs:15-25); `device` is not part of the model.
pub struct RCN<'a> {
    classes: usize,
    convpool_cfg: Vec<RCNLayer>,
    feedforward_cfg: Vec<usize>,
    layer_weights: Vec<Weights>,
    layer_bias: Vec<Bias>,
    scale_set: (f64, f64),

    training_path: &'a str,
    testing_path: &'a str,

    /// The device context, created at the first image (its shape is part of the context) and kept while the shape stays the
    /// same.  Behind a mutex because `classify` takes `&self` (rcn.rs:82) while every C-ABI call needs exclusive use of a context.
    #[serde(skip)]
    device: Mutex<Option<Device>>,
}

/// Weights tuple struct wrapped for serializing/deserializing (rcn.rs:28).
pub struct Weights(pub DMatrix<f64>);

/// Biases tuple struct wrapped for serializing/deserializing (rcn.rs:31).
pub struct Bias(pub DVector<f64>);

#[derive(Serialize, Deserialize)]
/// Characteristics of a layer (rcn.rs:35-38)
pub enum RCNLayer {
    Convolve2D(Padding),
    Pool2D(Pooling),
}

struct Device {
    ctx: sys::Context,
    shape: (usize, usize),
    /// the host copies (`layer_weights`, `layer_bias`, `scale_set`) have been pushed since they last changed
    params_pushed: bool,
}

/// One decoded data set: row-major luma bytes of every image back to back, class index per image (rcn.rs:377, 401).
struct ImageSet {
    pixels: Vec<u8>,
    labels: Vec<i32>,
    shape: (usize, usize),
}

impl<'a> RCN<'a> {
    /// Create new RCN instance (rcn.rs:58-75): stores the configuration; weights stay empty until `train`.
    pub fn new(
        classes: usize,
        convpool_cfg: Vec<RCNLayer>,
        feedforward_cfg: Vec<usize>,
        training_path: &'a str,
        testing_path: &'a str,
    ) -> Self {
        RCN {
            classes,
            convpool_cfg,
            feedforward_cfg,
            layer_weights: Vec::new(),
            layer_bias: Vec::new(),
            scale_set: (1_f64, 1_f64),
            training_path,
            testing_path,
            device: Mutex::new(None),
        }
    }

    /// Classify a given image and return the respective class index (rcn.rs:82-98): decode, grayscale, then ONE device call does
    /// flatten_feature_set, the standardisation with `scale_set`, `classify_test` and the arg-max (last maximum, like
    /// `max_by(total_cmp)`).
    pub fn classify(&self, img_path: &str) -> Result<usize, Box<dyn std::error::Error>> {
        let img = ImageReader::open(img_path)?.decode()?.grayscale();
        let (w, h, pixels) = crate::gray_bytes(&img).ok_or(crate::errors::InvalidGrayscaleImageError)?;
        let mut guard = self.device.lock().expect("device mutex poisoned");
        let dev = self.device_for(&mut *guard, (h, w));
        self.push_model(dev);
        Ok(dev.ctx.classify_image(&pixels)?)
    }

    /// Train the model (rcn.rs:126-167).
    ///
    /// # Arguments
    /// * `batch_size` - The number of samples processed before the model is updated
    /// * `epochs` - The number of total passes through the training set
    /// * `eta` - The learning rate
    /// * `class_size_limit` - A limiter on the number of samples to use per class
    ///
    pub fn train(
        &mut self,
        batch_size: usize,
        epochs: usize,
        eta: f64,
        training_class_size_limit: usize,
        testing_class_size_limit: usize,
    ) -> Result<(), ImageError> {
        let training = self.read_set(self.training_path, training_class_size_limit)?;
        let testing = self.read_set(self.testing_path, testing_class_size_limit)?;
        assert_eq!(training.shape, testing.shape, "training and testing images differ in size");

        let mut slot = self.device.lock().expect("device mutex poisoned").take();
        let dev = match &mut slot {
            Some(d) if d.shape == training.shape => d,
            other => other.insert(self.open_device(training.shape)),
        };
        // load_data x 2 (rcn.rs:134-137): features, gen_scales, standardise, one-hot -- resident in HBM from here on; scale_set
        // ends up holding the TEST set's statistics, as in the reference
        dev.ctx.load_data(0, &training.pixels, &training.labels).unwrap_or_else(|e| panic!("{}", e.message));
        self.scale_set = dev.ctx.load_data(1, &testing.pixels, &testing.labels).unwrap_or_else(|e| panic!("{}", e.message));

        if self.layer_weights.is_empty() {
            dev.ctx.init_params(0).unwrap_or_else(|e| panic!("{}", e.message)); // load_weights_and_bias, rcn.rs:139-141
        } else {
            Self::push_params(dev, &self.layer_weights, &self.layer_bias);      // resume from a deserialised model
        }
        dev.params_pushed = true;

        let mut order: Vec<i32> = (0..training.labels.len() as i32).collect();
        let test_len = testing.labels.len();
        for e in 0..epochs {
            order.shuffle(&mut rand::thread_rng());                                  // rcn.rs:146
            // for batch in training_set.chunks_exact(batch_size) { train_batch(batch, eta) }      rcn.rs:147-149
            dev.ctx.train_set_epoch(0, &order, batch_size, eta).unwrap_or_else(|e| panic!("{}", e.message));
            let accepted = dev.ctx.evaluate_set(1).unwrap_or_else(|e| panic!("{}", e.message)); // rcn.rs:152-157
            println!(
                "Epoch {}: {}/{} [{:.2}%]",
                e,
                accepted,
                test_len,
                accepted as f64 / test_len as f64 * 100_f64
            );
        }

        // the trained parameters come back into the serialisable fields
        let layers = dev.ctx.num_layers();
        self.layer_weights.clear();
        self.layer_bias.clear();
        for l in 0..layers {
            let (rows, cols) = dev.ctx.layer_dims(l);
            let (mut w, mut b) = (vec![0.0; rows * cols], vec![0.0; rows]);
            dev.ctx.get_params(l, &mut w, &mut b).unwrap_or_else(|e| panic!("{}", e.message));
            self.layer_weights.push(Weights(DMatrix::from_vec(rows, cols, w)));
            self.layer_bias.push(Bias(DVector::from_vec(b)));
        }
        *self.device.lock().expect("device mutex poisoned") = slot;
        Ok(())
    }

    // ------------------------------------------------------------------------------------------------ private

    fn layer_descriptors(&self) -> Vec<sys::rcn_hip_layer> {
        self.convpool_cfg
            .iter()
            .map(|l| match l {
                RCNLayer::Convolve2D(p) => sys::rcn_hip_layer { kind: sys::RCN_HIP_LAYER_CONVOLVE2D, arg: padding_tag(p) },
                RCNLayer::Pool2D(p) => sys::rcn_hip_layer { kind: sys::RCN_HIP_LAYER_POOL2D, arg: pooling_tag(p) },
            })
            .collect()
    }

    fn open_device(&self, shape: (usize, usize)) -> Device {
        // f64 on the device: the reference's own arithmetic type (rcn.rs:28,31,49); RCN_HIP_F32 is the faster context when the
        // caller accepts the f32 tolerances of DESIGN.md §5
        let ctx = sys::Context::new(self.classes, &self.layer_descriptors(), &self.feedforward_cfg, shape.0, shape.1, sys::RCN_HIP_F64, 0)
            .unwrap_or_else(|e| panic!("{}", e.message));
        Device { ctx, shape, params_pushed: false }
    }

    fn device_for<'g>(&self, guard: &'g mut Option<Device>, shape: (usize, usize)) -> &'g mut Device {
        if guard.as_ref().map(|d| d.shape) != Some(shape) {
            *guard = Some(self.open_device(shape));
        }
        guard.as_mut().unwrap()
    }

    fn push_params(dev: &mut Device, weights: &[Weights], bias: &[Bias]) {
        for (l, (w, b)) in weights.iter().zip(bias.iter()).enumerate() {
            // DMatrix::as_slice() is column-major: exactly the layout rcn_hip_set_params takes
            dev.ctx.set_params(l, w.0.as_slice(), b.0.as_slice()).unwrap_or_else(|e| panic!("{}", e.message));
        }
    }

    /// Model state of a freshly deserialised `RCN` (backend/src/main.rs:64-70) -> device, once.
    fn push_model(&self, dev: &mut Device) {
        if !dev.params_pushed {
            Self::push_params(dev, &self.layer_weights, &self.layer_bias);
            dev.ctx.set_scale(self.scale_set.0, self.scale_set.1).unwrap_or_else(|e| panic!("{}", e.message));
            dev.params_pushed = true;
        }
    }

    /// The file half of `load_data` (rcn.rs:367-404): class directories in lexicographic order (= class index), `class_size_limit`
    /// files drawn per class without replacement, decoded to grayscale.
    fn read_set(&self, path: &str, class_size_limit: usize) -> Result<ImageSet, ImageError> {
        let mut classes: Vec<PathBuf> = fs::read_dir(path)
            .unwrap_or_else(|e| panic!("could not read {path}: {e}"))
            .map(|entry| entry.expect("directory entry").path())
            .collect();
        classes.sort();
        let mut set = ImageSet { pixels: Vec::new(), labels: Vec::new(), shape: (0, 0) };
        let mut rng = rand::thread_rng();
        for (class_index, dir) in classes.iter().enumerate() {
            let files: Vec<PathBuf> = fs::read_dir(dir)
                .unwrap_or_else(|e| panic!("could not read {}: {e}", dir.display()))
                .map(|entry| entry.expect("directory entry").path())
                .collect();
            if class_size_limit > files.len() {
                panic!(
                    "provided class_size_limit for {} too large! expected {} <= {}",
                    path,
                    class_size_limit,
                    files.len()
                );
            }
            for file in files.choose_multiple(&mut rng, class_size_limit) {
                let img = ImageReader::open(file)?.decode()?.grayscale();
                let (w, h, bytes) = crate::gray_bytes(&img).expect("grayscale() yields Luma8 / LumaA8");
                if set.labels.is_empty() {
                    set.shape = (h, w);
                }
                assert_eq!(set.shape, (h, w), "images of one data set must share one size");
                set.pixels.extend_from_slice(&bytes);
                set.labels.push(class_index as i32);
            }
        }
        Ok(set)
    }
}
