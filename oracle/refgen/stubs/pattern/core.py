panel_spec_template = {}
