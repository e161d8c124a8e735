"""Import-path compatibility with the reference package layout: a script written against
`vibevoice.modular.modeling_vibevoice_inference` / `vibevoice.processor.vibevoice_processor` (e.g. the reference's
demo/inference_from_file.py) runs on the MI355X engine unchanged."""
