"""Drop-in surface of the reference package layout (`src.models`, `src.training`) backed by the
MI355X engine in `graph-gpt_amd/`."""
