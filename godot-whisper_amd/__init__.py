"""MI355X-native Whisper inference backend behind the godot-whisper host boundary.

Directory name carries a hyphen (repo convention); import it through
`__graft_entry__.load_package()` which registers it as `godot_whisper_amd`.
"""
from . import abi, synth, host, runtime, shard  # noqa: F401
