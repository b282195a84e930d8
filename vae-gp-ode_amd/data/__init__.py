"""Data path of the reference (experiments/data/): rotating-MNIST loaders with the reference's function names, plus
``ResidentLoader`` -- the form this build trains from (the whole set lives in HBM; a minibatch is one device-side gather)."""
