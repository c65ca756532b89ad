"""import-only stand-in (build container): GenericNNetWrapper.py:20-21 imports onnx / onnxruntime at module level; the CPU
training path exercised by tools/gen_train_golden.py never calls into them."""
