"""
gen_adversarial_amd — MI355X-native purification-under-attack hot path (NVAE purify + classifier, forward and
input-gradient) behind the defender API of SerezD/gen_adversarial.  The HIP/C-ABI library is loaded lazily by
`gen_adversarial_amd._lib`; there is no CPU fallback on the product path.
"""
__version__ = '0.1.0'
