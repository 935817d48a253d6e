pub mod kernel;
pub mod serialization;
