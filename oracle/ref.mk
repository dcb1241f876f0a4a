# oracle/ref.mk -- TEST INFRASTRUCTURE ONLY.
# Compiles the reference's own hifiasm sources *where they lie* under
# $(REF) (read-only) into oracle/_ref/ (git-ignored, travels with gpurun).
# Nothing is copied into this repo; no stand-in headers are written.
#   make -f oracle/ref.mk            # builds hifiasm-0.14, hifiasm-0.16.1, harnesses
REF      ?= /root/reference
OUT      := oracle/_ref
HA14     := $(REF)/software/hifiasm-0.14
HA16     := $(REF)/software/hifiasm-0.16.1
CXXFLAGS := -O3 -msse4.2 -mpopcnt -fomit-frame-pointer -w
LIBS     := -lz -lpthread -lm

HA14_SRC := $(filter-out $(HA14)/main.cpp $(HA14)/Output.cpp,$(wildcard $(HA14)/*.cpp))
HA14_OBJ := $(patsubst $(HA14)/%.cpp,$(OUT)/ha14/%.o,$(HA14_SRC)) $(OUT)/ha14/ksw2_extz2_sse.o
HA16_SRC := $(filter-out $(HA16)/main.cpp $(HA16)/Output.cpp,$(wildcard $(HA16)/*.cpp))
HA16_OBJ := $(patsubst $(HA16)/%.cpp,$(OUT)/ha16/%.o,$(HA16_SRC)) $(OUT)/ha16/ksw2_extz2_sse.o

.PHONY: all ha14 ha16 harness
all: ha14 ha16 harness
ha14: $(OUT)/hifiasm-0.14
ha16: $(OUT)/hifiasm-0.16.1
harness: $(OUT)/ha14_kernels $(OUT)/hifiasm_trace

$(OUT)/ha14/%.o: $(HA14)/%.cpp
	@mkdir -p $(@D)
	g++ -c $(CXXFLAGS) $< -o $@
$(OUT)/ha14/%.o: $(HA14)/%.c
	@mkdir -p $(@D)
	gcc -c $(CXXFLAGS) $< -o $@
$(OUT)/ha16/%.o: $(HA16)/%.cpp
	@mkdir -p $(@D)
	g++ -c $(CXXFLAGS) $< -o $@
$(OUT)/ha16/%.o: $(HA16)/%.c
	@mkdir -p $(@D)
	gcc -c $(CXXFLAGS) $< -o $@

$(OUT)/hifiasm-0.14: $(HA14_OBJ) $(OUT)/ha14/main.o
	g++ $(CXXFLAGS) $^ -o $@ $(LIBS)
$(OUT)/hifiasm-0.16.1: $(HA16_OBJ) $(OUT)/ha16/main.o
	g++ $(CXXFLAGS) $^ -o $@ $(LIBS)

# known-answer harness: our driver (oracle/ref_harness.cpp) includes the
# reference's Correct.cpp as a translation unit to reach its inline K5/K6
# routines; links against the other reference objects.
$(OUT)/ha14_kernels: oracle/ref_harness.cpp $(filter-out $(OUT)/ha14/Correct.o,$(HA14_OBJ))
	g++ $(CXXFLAGS) -I$(HA14) -DREF_CORRECT_CPP='"$(HA14)/Correct.cpp"' $^ -o $@ $(LIBS)

# diagnosis: the reference binary with a trace of its graph steps (oracle/ref_graph_trace.cpp includes the reference's Overlaps.cpp
# as a translation unit and replays clean_graph's call sequence with a dump after each step)
$(OUT)/hifiasm_trace: oracle/ref_graph_trace.cpp $(filter-out $(OUT)/ha14/Overlaps.o,$(HA14_OBJ)) $(OUT)/ha14/main.o
	g++ -O1 -msse4.2 -mpopcnt -w -I$(HA14) -DREF_OVERLAPS_CPP='"$(HA14)/Overlaps.cpp"' $^ -o $@ $(LIBS)
