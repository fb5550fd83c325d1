"""advanced_rag — host-side mirror of the reference package's hot-path API,
backed by libhbmrag (HIP kernels for gfx950)."""
