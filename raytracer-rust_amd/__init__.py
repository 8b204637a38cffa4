"""raytracer-rust_amd -- MI355X-native render loop for jackra1n/raytracer-rust.

The directory name carries a hyphen, so import it with
    importlib.import_module("raytracer-rust_amd")
Sub-modules: abi (ctypes mirror of include/mi355rt.h), build (hipcc / g++ recipes),
device (binding of libmi355rt.so, the HIP path), host (binding of libmi355rt_host.so).
Nothing here falls back to a CPU renderer: device.* raises if the HIP library or a GPU is missing.
"""
__all__ = ["abi"]
