"""Reference import paths (`training.*`) resolved to the MI355X implementation in adam-dehaze_amd/."""
