"""CPU ORACLE, part 4 (test infrastructure, NOT the product): the two hash functions behind TensorFlow's
``sparse_cross_hashed`` -- what ``tf.feature_column.crossed_column`` (WideNDeep.py:72-73) computes its buckets with.

Third-party owner of the arithmetic: TensorFlow (core/platform/fingerprint.h ``FingerprintCat64``; core/kernels/
sparse_cross_op.cc ``HashCrosser``) on top of Google FarmHash (``farmhash::Fingerprint64`` = ``farmhashna::Hash64``),
neither vendored in /root/reference.  Restated here from the published algorithms in plain Python integers.

PINNED: unlike the rest of the oracle this piece has golden vectors -- TensorFlow's own kernel test
python/kernel_tests/sparse_cross_op_test.py crosses the strings 'batch1-FC1-F1' x 'batch1-FC2-F1' x 'batch1-FC3-F1' and
asserts ("Check actual hashed output to prevent unintentional hashing changes")
    test_hashed_zero_bucket_no_hash_key   num_buckets 0, default key 0xDECAFCAFFE      -> 1971693436396284976
    test_hashed_zero_bucket               num_buckets 0, key 0xDECAFCAFFE + 1          -> 4847552627144134031
    test_hashed_no_hash_key               num_buckets 100, default key                 -> 83
    test_hashed_output                    num_buckets 100, key 0xDECAFCAFFE + 1        -> 31
(num_buckets 0 means ``h mod int64max``).  tests/test_farmhash_pins.py reproduces all four through THESE functions: two
independent 64-bit matches cannot happen by accident, and 83 = (1971693436396284976 + 2^63 - 1) mod 100 shows the raw
hash is >= 2^63, i.e. the modulo is taken on the UNSIGNED value.  That pins ``FingerprintCat64``, the default hash key, the
chaining order and the modulo -- everything the reference's int64 cross (movieId x userRatedMovie1) uses; the only step
a string-free cross does not share with the vectors is that an int64 feature enters as its value (sparse_cross_op.cc
``SparseTensorColumn<int64>::Feature`` returns ``values(start + n)``) instead of ``Fingerprint64(string)``.
"""
M64 = (1 << 64) - 1
DEFAULT_HASH_KEY = 0xDECAFCAFFE            # python/ops/sparse_ops.py _DEFAULT_HASH_KEY
INT64_MAX = (1 << 63) - 1

_K0 = 0xc3a5c85c97cb3127
_K1 = 0xb492b66fbe98f273
_K2 = 0x9ae16a3b2f90404f
_KMUL = 0xc6a4a7935bd1e995


def _shift_mix(v: int) -> int:
    return v ^ (v >> 47)


def _rot(v: int, s: int) -> int:
    return v if s == 0 else ((v >> s) | (v << (64 - s))) & M64


def _f64(b: bytes, i: int) -> int:
    return int.from_bytes(b[i:i + 8], "little")


def _f32(b: bytes, i: int) -> int:
    return int.from_bytes(b[i:i + 4], "little")


def _hash_len16(u: int, v: int, mul: int) -> int:
    a = ((u ^ v) * mul) & M64
    a ^= a >> 47
    b = ((v ^ a) * mul) & M64
    b ^= b >> 47
    return (b * mul) & M64


def _hash_len0to16(s: bytes) -> int:
    n = len(s)
    if n >= 8:
        mul = (_K2 + n * 2) & M64
        a = (_f64(s, 0) + _K2) & M64
        b = _f64(s, n - 8)
        c = (_rot(b, 37) * mul + a) & M64
        d = ((_rot(a, 25) + b) * mul) & M64
        return _hash_len16(c, d, mul)
    if n >= 4:
        mul = (_K2 + n * 2) & M64
        return _hash_len16((n + (_f32(s, 0) << 3)) & M64, _f32(s, n - 4), mul)
    if n > 0:
        y = (s[0] + (s[n >> 1] << 8)) & 0xffffffff
        z = (n + (s[n - 1] << 2)) & 0xffffffff
        return (_shift_mix((y * _K2 ^ z * _K0) & M64) * _K2) & M64
    return _K2


def _hash_len17to32(s: bytes) -> int:
    n = len(s)
    mul = (_K2 + n * 2) & M64
    a = (_f64(s, 0) * _K1) & M64
    b = _f64(s, 8)
    c = (_f64(s, n - 8) * mul) & M64
    d = (_f64(s, n - 16) * _K2) & M64
    return _hash_len16((_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64, (a + _rot((b + _K2) & M64, 18) + c) & M64, mul)


def fingerprint64(s: bytes) -> int:
    """farmhash::Fingerprint64 for strings of at most 32 bytes (all this repository hashes: feature values)."""
    if len(s) <= 16:
        return _hash_len0to16(s)
    if len(s) <= 32:
        return _hash_len17to32(s)
    raise NotImplementedError("fingerprint64: strings longer than 32 bytes are not needed here")


def fingerprint_cat64(fp0: int, fp1: int) -> int:
    """core/platform/fingerprint.h FingerprintCat64."""
    r = (fp0 ^ _KMUL) & M64
    r ^= (_shift_mix((fp1 * _KMUL) & M64) * _KMUL) & M64
    r = (r * _KMUL) & M64
    r = (_shift_mix(r) * _KMUL) & M64
    return _shift_mix(r)


def cross_hashed(features, num_buckets: int = 0, hash_key: int = DEFAULT_HASH_KEY) -> int:
    """One example of ``sparse_cross_hashed``: ``features`` in column order, ints (used as they are) or bytes/str
    (fingerprinted).  sparse_cross_op.cc HashCrosser::Generate."""
    h = hash_key & M64
    for f in features:
        if isinstance(f, str):
            f = f.encode("utf-8")
        fp = fingerprint64(f) if isinstance(f, (bytes, bytearray)) else int(f) & M64
        h = fingerprint_cat64(h, fp)
    return h % num_buckets if num_buckets > 0 else h % INT64_MAX
