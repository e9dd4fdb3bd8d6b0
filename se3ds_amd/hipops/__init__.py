"""autograd-facing wrappers of the libse3ds_hip.so network kernels."""
