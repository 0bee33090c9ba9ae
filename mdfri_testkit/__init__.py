"""Test and benchmark support that is NOT part of the product package `mDeepFRI`: seeded synthetic workloads and weights
(`synthetic`) and the minimal ONNX exporter (`onnx_writer`) that the reader tests, the validation kit and
bench.py's optional onnxruntime leg use.  Imported by tests/, tools/, bench.py and __graft_entry__.smoke() only."""
