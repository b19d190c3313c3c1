"""replay_cql_amd -- MI355X-native CQL recommender hot path behind the RePlay Recommender API.

Only what the path needs: csrc/ (HIP kernels + C ABI, built into libcqlrec.so), _native (ctypes binding),
core (device driver).  The HIP library is loaded lazily; nothing here falls back to a CPU implementation."""
__version__ = "0.1.0"
