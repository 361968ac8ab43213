"""MI355X-native batched FJSP scheduling environment (gfx950 HIP kernels behind
the reference's Gym-style reset()/step() surface).  See DESIGN.md."""

__version__ = "0.1.0"
