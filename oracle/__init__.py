"""CPU oracle package — test infrastructure only (see whisper_ref.c header)."""
