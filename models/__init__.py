"""Reference import paths (`models.*`) resolved to the MI355X implementation in adam-dehaze_amd/."""
