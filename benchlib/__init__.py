"""Pieces of bench.py (the driver's entry point at the repo root): the synthetic workload, the rocprofv3 PMC child passes, the
launcher, one rank's job, the roofline block, the side legs."""
