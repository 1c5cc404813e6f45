"""Prompt-suffix vocabularies of the generation loop (the reference's
prompts_engineering/__init__.py constants; they enter output FILE NAMES, so they are part of
the output contract).  The 100-line GPT prompt files are dataset-side data and are read from
the path given by Settings.PROMPTS_FILE (default: the reference checkout's
prompts_engineering/gpt_prompts/<meta_class>-100-gpt_v1.txt)."""

_PAINTERS = ("van gogh", "monet", "picasso", "da vinci", "michelangelo", "rembrandt", "raphael", "vermeer", "degas", "klimt")
ARTISTIC_PROMPTS = [f"a painting of {name}" for name in _PAINTERS]

IMAGE_VARIATIONS_PROMPTS = [
    "High-Speed", "Lens Flare", "HDR (High Dynamic Range)", "Fish-Eye Lens", "Black and White", "Long Exposure", "Macro",
    "Panoramic", "Tilt-Shift", "Infrared", "Bokeh", "Time-Lapse", "Underwater", "Double Exposure", "Sepia Tone",
    "Vintage Look", "Solarized", "Low Light", "Motion Blur", "Cross Processed",
]
